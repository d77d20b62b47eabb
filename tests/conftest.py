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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How much of each compared tensor was actually held to the parity bar (tests/util.py ACCOUNTING):
    total pixels / rows, those on the strict bar, those on the flip-sized bound next to a fragile
    evaluation, those exempt.  Also written to gpurun_out/parity_accounting.json when that directory
    exists (the GPU box), so the numbers can be quoted."""
    try:
        from tests import util
    except Exception:
        return
    acc = util.ACCOUNTING
    if not acc:
        return
    tr = terminalreporter
    if util.PSNR_LOG:
        tr.section("PSNR of the HIP renders against the oracle's (reference metric, src/evaluation/metrics.py:13-20)")
        for r in util.PSNR_LOG:
            tr.write_line(f"{r['what']:64s} {r['db']:8.2f} dB")
    tr.section("parity accounting (what was held to the bar)")
    worst = {}
    for a in acc:
        key = (a["kind"], a["what"].split("[")[0])
        w = worst.setdefault(key, dict(n=0, total=0, strict=0, bounded=0, exempt=0, min_frac=1.0, worst=0.0, row=None))
        w["n"] += 1; w["total"] += a["total"]; w["strict"] += a["held_to_bar"]; w["bounded"] += a["flip_bounded"]
        w["exempt"] += a["exempt"]; w["worst"] = max(w["worst"], a["worst_strict_err"] / a["scale"])
        w["min_frac"] = min(w["min_frac"], a["held_to_bar"] / max(a["total"], 1))
        if "worst_clean_row_mixed" in a:
            w["row"] = max(w["row"] or 0.0, a["worst_clean_row_mixed"])
    for (kind, what), w in sorted(worst.items()):
        # images: largest error among the pixels on the bar; gradients: largest error of a CLEAN row (no fragile
        # evaluation nearby) relative to the tensor's scale, and relative to max(1, |row|) ("row-mixed")
        tail = f"worst err/scale {w['worst']:.2e}" + (f"  worst row-mixed {w['row']:.2e}" if w["row"] is not None else "")
        tr.write_line(f"{kind:5s} {what:28s} comparisons {w['n']:4d}  elements {w['total']:9d}  on the bar {w['strict']:9d} "
                      f"({100.0 * w['strict'] / max(w['total'], 1):6.2f} %, min {100.0 * w['min_frac']:6.2f} %)  "
                      f"flip-bounded {w['bounded']:6d}  exempt {w['exempt']:5d}  {tail}")
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "parity_accounting.json"), "w") as f:
            json.dump(acc + [dict(kind="psnr", **r) for r in util.PSNR_LOG], f)
