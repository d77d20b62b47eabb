import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The in-tree shared objects are build artefacts (git-ignored): (re)build them when absent so a
    # fresh checkout can run the suite.  hipcc cross-compiles gfx950 without a GPU; on a box
    # without hipcc the prebuilt .so that travelled with the tree is used as is.
    import shutil
    from latentsplat_amd import _lib
    if not os.path.exists(_lib.so_path()) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        _lib.build()


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no ROCm device is visible")
    from latentsplat_amd import _lib
    _lib.load()  # fail loudly if the extension is missing
    return torch.device("cuda:0")
