"""A short, fixed-seed leg of the randomised parity sweep (tests/fuzz_parity.py) inside the suite: 48 random shapes from
the corners the kernels switch on, forward lists bit for bit and images / gradients at 1e-4 against the CPU oracle.  The
long sweeps (thousands of cases per seed) are run by hand on the GPU box: profiles/r04_fuzz_parity.md."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_short_sweep(hip_device):
    from tests import fuzz_parity
    res = fuzz_parity.sweep(hip_device, seed=20260927, seconds=120.0, max_cases=48)
    assert res["cases"] == 48 or res["seconds"] >= 120.0
    assert not res["failures"], res["failures"][:3]
