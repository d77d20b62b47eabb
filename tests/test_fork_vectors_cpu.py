"""CPU half of the fork-parity plumbing:
  * the dump script + replay round-trip, exercised with the CPU oracle standing in for the fork
    (keeps the file format and the consumer honest without a GPU);
  * when tests/golden/fork_*.npz exist (dumped from the REAL fork on a CUDA machine): the ORACLE is
    checked against them — that is what would turn 'parity unpinned' into 'pinned'."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from tests import fork_vectors as fv  # noqa: E402


def test_dump_script_round_trip_with_oracle(tmp_path):
    import dump_fork_vectors as dump
    dump.main(["--out", str(tmp_path), "--module", "oracle.oracle", "--device", "cpu", "--prefix", "selftest_",
               "--only", "fork_cfg1_feat4,fork_probe_layers,fork_probe_sh_axes"])
    mod = importlib.import_module("oracle.oracle")
    files = sorted(os.listdir(tmp_path))
    assert files == ["selftest_fork_cfg1_feat4.npz", "selftest_fork_probe_layers.npz", "selftest_fork_probe_sh_axes.npz"]
    for f in files:
        assert os.path.getsize(tmp_path / f) < 1 << 20
        res = fv.replay(str(tmp_path / f), mod, "cpu")
        assert any(k.endswith("out_mask") for k in res) and any("grad_means3D" in k for k in res)
        assert fv.compare(res, 1e-6, 1e-6) == [], f


def test_every_case_is_dumpable_with_the_oracle(tmp_path):
    """All six cases go through (shapes, SH degree 4 payload, ragged image, 8 channels)."""
    import dump_fork_vectors as dump
    dump.main(["--out", str(tmp_path), "--module", "oracle.oracle", "--device", "cpu"])
    assert len(os.listdir(tmp_path)) == len(dump.cases())


@pytest.mark.skipif(not fv.fork_files(), reason=fv.SKIP_REASON)
def test_oracle_matches_the_real_fork():
    mod = importlib.import_module("oracle.oracle")
    problems = []
    for path in fv.fork_files():
        problems += [f"{os.path.basename(path)}: {m}" for m in fv.compare(fv.replay(path, mod, "cpu"), image_outliers=0.002)]
    assert not problems, "\n".join(problems)
