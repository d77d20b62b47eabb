"""Latency mode (lsr_forward_nosync): the forward without the host read-back of the pair count
(upstream's blocking `num_rendered` copy), its overflow behaviour,
the backward on top of it, and hipGraph capture / replay through torch.cuda.graph."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _inputs(dev, G=20000, V=2, size=128, seed=7, sh=2):
    sc = util.make_scene(G, image_size=size, views=V, color_sh_degree=sh, feature_channels=4, feature_sh_degree=0, seed=seed)
    bi = util.boundary_inputs(sc, size, size, bg=(0.1, 0.2, 0.3))
    t = {k: bi[k].to(dev) for k in ("means", "cov6", "opac", "shs", "features")}
    return bi, util.view_table(bi, dev), t, size


def _render(views, t, size, deg, **kw):
    from latentsplat_amd.rasterizer import rasterize_views
    return rasterize_views(views, size, size, deg, t["means"], t["cov6"], t["opac"], shs=t["shs"], features=t["features"], **kw)


def test_nosync_forward_equals_synchronous_forward(hip_device):
    from latentsplat_amd.rasterizer import last_forward_status
    bi, views, t, size = _inputs(hip_device)
    ref = _render(views, t, size, 2)
    st = last_forward_status()
    assert st["num_pairs"] > 20000 and not st["overflow"]
    for hint in (64, st["max_tile_pairs"], 16384):        # the hint only picks the sort variant
        out = _render(views, t, size, 2, pair_capacity=int(1.5 * st["num_pairs"]), max_tile_hint=hint)
        st2 = last_forward_status()
        assert st2 == st
        for a, b in zip(out, ref):
            assert torch.equal(a, b)


def test_nosync_overflow_is_flagged_and_harmless(hip_device):
    from latentsplat_amd.rasterizer import last_forward_status
    bi, views, t, size = _inputs(hip_device)
    ref = _render(views, t, size, 2)
    P = last_forward_status()["num_pairs"]
    out = _render(views, t, size, 2, pair_capacity=P // 2, max_tile_hint=1024)
    st = last_forward_status()
    assert st["overflow"] and st["num_pairs"] == P           # the true count is still reported
    assert all(torch.isfinite(x).all() for x in out[:4])
    assert not torch.equal(out[2], ref[2])                   # truncated lists: the result is NOT to be used
    again = _render(views, t, size, 2, pair_capacity=P + 1, max_tile_hint=1024)     # exactly enough room
    assert not last_forward_status()["overflow"] and torch.equal(again[0], ref[0])


def test_backward_after_nosync_forward(hip_device):
    from latentsplat_amd.rasterizer import last_forward_status
    bi, views, t, size = _inputs(hip_device, G=8000, size=64)
    g = torch.randn((2, 3, size, size), generator=torch.Generator().manual_seed(2)).to(hip_device)

    def grads(**kw):
        leaf = {k: v.clone().requires_grad_(True) for k, v in t.items()}
        out = _render(views, leaf, size, 2, **kw)
        ((out[0] * g).sum() + (out[1] ** 2).sum()).backward()
        return {k: v.grad for k, v in leaf.items()}

    ref = grads()
    P = last_forward_status()["num_pairs"]
    got = grads(pair_capacity=2 * P, max_tile_hint=512)
    for k in ref:
        scale = max(1.0, float(ref[k].abs().max()))
        assert float((got[k] - ref[k]).abs().max()) <= 1e-5 * scale, k        # atomic-add ordering only


def test_graph_capture_and_replay(hip_device):
    """The no-sync forward is a pure launch sequence: captured once, replayed with new scene contents
    in the static input tensors, equal to the eager result."""
    from latentsplat_amd.rasterizer import last_forward_status
    dev = hip_device
    bi, views, t, size = _inputs(dev, G=30000, V=1, size=128, seed=11)
    _render(views, t, size, 2)
    P = last_forward_status()["num_pairs"]
    static = {k: v.clone() for k, v in t.items()}
    kw = dict(pair_capacity=2 * P, max_tile_hint=2048)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            _render(views, static, size, 2, **kw)           # warm-up on the capture stream
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = _render(views, static, size, 2, **kw)
    graph.replay()
    torch.cuda.synchronize(dev)
    eager = _render(views, t, size, 2)
    for a, b in zip(out, eager):
        assert torch.equal(a, b)
    # new scene contents, same shapes: refill the static tensors and replay
    bi2, views2, t2, _ = _inputs(dev, G=30000, V=1, size=128, seed=12)
    for k in static:
        static[k].copy_(t2[k])
    graph.replay()
    torch.cuda.synchronize(dev)
    assert not last_forward_status()["overflow"]
    eager2 = _render(views, t2, size, 2)                     # same cameras (the view table was captured by address)
    for a, b in zip(out, eager2):
        assert torch.equal(a, b)
    o = util.oracle_forward(bi2, 0)
    util.assert_close_except_fragile(out[0][0].cpu().numpy(), o["color"], o, 1e-4, "graph replay colour")


def test_graph_capture_of_forward_and_backward(hip_device):
    """Round 6: a training-mode no-sync forward forks the gradient-workspace clear onto the library's side stream (beside the
    compositing kernel) and joins it back inside the call, and orders the backward's work items with one more small kernel:
    a hipGraph capture of forward + backward records all of it and replays to the eager gradients (float-atomic order only)."""
    from latentsplat_amd.rasterizer import last_forward_status
    dev = hip_device
    bi, views, t, size = _inputs(dev, G=8000, size=64)
    g = torch.randn((2, 3, size, size), generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        _render(views, t, size, 2)
    P = last_forward_status()["num_pairs"]
    kw = dict(pair_capacity=2 * P, max_tile_hint=2048)
    leaf = {k: v.clone().requires_grad_(True) for k, v in t.items()}

    def step():
        out = _render(views, leaf, size, 2, **kw)
        ((out[0] * g).sum() + (out[1] ** 2).sum()).backward()

    step()                                   # eager reference
    ref = {k: v.grad.clone() for k, v in leaf.items()}
    side = torch.cuda.Stream(dev)            # warm-up on a side stream, as torch.cuda.graph asks for
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for v in leaf.values():
            v.grad = None
        step()
    torch.cuda.current_stream(dev).wait_stream(side)
    for v in leaf.values():
        v.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(dev)
    for k, v in leaf.items():
        scale = max(1.0, float(ref[k].abs().max()))
        assert torch.isfinite(v.grad).all(), k
        assert float((v.grad - ref[k]).abs().max()) <= 1e-5 * scale, k


def test_deterministic_backward_mode(hip_device):
    """LSR_DETERMINISTIC=1 (read once per process, hence a subprocess): the cross-tile gradient sums use
    64-bit fixed-point atomics, so two runs give bitwise identical gradients, which agree with the
    default (float atomic) mode to the fixed-point resolution."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch, hashlib
sys.path.insert(0, %r)
from tests.test_latency_gpu import _inputs, _render
dev = torch.device("cuda:0")
bi, views, t, size = _inputs(dev, G=8000, size=64)
g = torch.randn((2, 3, size, size), generator=torch.Generator().manual_seed(2)).to(dev)
def grads():
    leaf = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    out = _render(views, leaf, size, 2)
    ((out[0] * g).sum() + (out[1] ** 2).sum()).backward()
    return torch.cat([v.grad.reshape(-1) for v in leaf.values()])
a, b = grads(), grads()
print("EQUAL", bool(torch.equal(a, b)), hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest(), float(a.abs().max()))
torch.save(a.cpu(), sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for det in ("1", "0"):
        path = f"/tmp/lsr_det_{det}.pt"
        env = dict(os.environ, LSR_DETERMINISTIC=det)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[det] = (r.stdout.strip().splitlines()[-1], torch.load(path))
    assert outs["1"][0].startswith("EQUAL True"), outs["1"][0]          # bitwise reproducible
    a, b = outs["1"][1], outs["0"][1]
    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


def test_launch_structure_knobs_do_not_change_results(hip_device):
    """Where the tile scan runs (folded into the last workgroup of k_preprocess or as a kernel of its own:
    LSR_FOLD_SCAN), how the host waits for the pair count (polling the mapped words or sleeping on the event:
    LSR_HOST_POLL), the order in which the per-tile sort visits the tiles (LSR_SORT_LPT) and whether the forward
    compositing kernel renders a half tile with one wave (two pixels per lane) or with two waves (one row of sub-blocks
    each, one pixel per lane: LSR_FWD_ROWS, chosen by the item count otherwise) and between how many waves the backward splits
    a half-tile list (LSR_BWD_PARTS), and the issue priorities the compositing waves give themselves (LSR_BWD_PRIO_PCT / LSR_FWD_PRIO_PCT) only change WHEN and WHERE kernels run: images, the per-pixel workspaces the backward reads and the gradients must be bitwise identical (knobs
    are read once per process, hence subprocesses), in the synchronous and the no-sync forward, under back-to-back
    calls and inside a captured hipGraph."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, torch, hashlib
os.environ["LSR_DETERMINISTIC"] = "1"      # order-independent backward sums: gradients comparable bit for bit
sys.path.insert(0, %r)
from tests.test_latency_gpu import _inputs
from latentsplat_amd.rasterizer import rasterize_views
dev = torch.device("cuda:0")
bi, views, t, size = _inputs(dev, G=30000, V=8, size=96, sh=2)
h = hashlib.sha1()
def call(**kw):
    leaves = [t[k].clone().requires_grad_(True) for k in ("means", "cov6", "opac", "shs", "features")]
    out = rasterize_views(views, size, size, 2, leaves[0], leaves[1], leaves[2], shs=leaves[3], features=leaves[4], **kw)
    return out, leaves
for kw in ({}, dict(pair_capacity=600000, max_tile_hint=2048)):
    for _ in range(3):
        out, leaves = call(**kw)
    (out[0].sum() * 0.5 + (out[1] * out[1]).sum() + out[2].sum() + out[3].sum()).backward()
    for o in list(out[:4]) + [l.grad for l in leaves]:
        h.update(o.detach().cpu().numpy().tobytes())
# the no-sync launch sequence (side-stream fork / join included) captured once and replayed
with torch.no_grad():
    kw = dict(pair_capacity=600000, max_tile_hint=2048)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        rasterize_views(views, size, size, 2, t["means"], t["cov6"], t["opac"], shs=t["shs"], features=t["features"], **kw)
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gout = rasterize_views(views, size, size, 2, t["means"], t["cov6"], t["opac"], shs=t["shs"], features=t["features"], **kw)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for o in gout[:4]:
        h.update(o.cpu().numpy().tobytes())
print("HASH", h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    # (LSR_FWD_QUAD=0 throughout: the sub-block-item kernel this small shape would otherwise get sums its colour in a
    # different order — test_subblock_items_forward_matches_the_serial_kernels holds it to its own contract)
    variants = {"default": {}, "scan_kernel": dict(LSR_FOLD_SCAN="0"), "event_wait": dict(LSR_HOST_POLL="0"),
                "scan_kernel_event_wait": dict(LSR_FOLD_SCAN="0", LSR_HOST_POLL="0"),
                "sort_natural_order": dict(LSR_SORT_LPT="0"),
                # round 5: the projection kernel writes the sort keys into fixed-capacity tile segments itself (default) or
                # leaves them to k_scatter behind the tile scan; a small segment capacity makes every call of this scene
                # overflow — the synchronous forward then re-runs itself on the two-phase path
                "two_phase_binning": dict(LSR_SEGMENTS="0"),
                "fwd_row_items": dict(LSR_FWD_ROWS="1"), "fwd_half_tile_items": dict(LSR_FWD_ROWS="0"),
                # the backward of this small shape splits every half-tile list between four waves (each walks the entries in
                # front of its share for the per-pixel state only): the same gradient records as one wave per list
                "bwd_unsplit_lists": dict(LSR_BWD_PARTS="0"), "bwd_eight_parts": dict(LSR_BWD_PARTS="3"),
                # round 5: the compositing waves change their issue priority with their progress through an item (on for some
                # instances by default): forced off, and forced on for every instance of the backward and of the half-tile forward
                "no_progress_priority": dict(LSR_BWD_PRIO_PCT="0", LSR_FWD_PRIO_PCT="0"),
                "progress_priority_everywhere": dict(LSR_BWD_PRIO_PCT="40", LSR_FWD_PRIO_PCT="30", LSR_FWD_ROWS="0"),
                "progress_priority_unsplit_bwd": dict(LSR_BWD_PRIO_PCT="25", LSR_BWD_PARTS="0")}
    for name, extra in variants.items():
        # (LSR_FWD_RECORD=0 throughout: the half-tile forward kernel narrows the render lists to the sub-blocks an entry
        # contributed to when a backward follows, the row / sub-block kernels do not — the backward's float sums then
        # group differently; test_forward_for_backward_narrows_the_render_lists_losslessly holds that to its own contract)
        # (LSR_BWD_REV=0 throughout: which items the backward walks back to front depends on flags that only some forward
        # kernels leave behind — round 6, tests/test_steep_alpha_gpu.py — and the two walk orders sum differently; the
        # back-to-front walk gets its own two variants below)
        env = dict(os.environ, LSR_FWD_QUAD="0", LSR_FWD_RECORD="0", LSR_BWD_REV="0", **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        digests[name] = r.stdout.strip().splitlines()[-1]
    assert digests["default"].startswith("HASH ") and len(set(digests.values())) == 1, digests
    # round 6: every item walked back to front — split between eight waves, four (the default here) or not at all: the
    # same gradient records bit for bit (a part walks the batches BEHIND its share for the per-pixel state only)
    rev = {}
    for name, extra in {"rev_default": {}, "rev_unsplit": dict(LSR_BWD_PARTS="0"), "rev_eight_parts": dict(LSR_BWD_PARTS="3")}.items():
        env = dict(os.environ, LSR_FWD_QUAD="0", LSR_FWD_RECORD="0", LSR_BWD_REV="1", **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        rev[name] = r.stdout.strip().splitlines()[-1]
    assert rev["rev_default"].startswith("HASH ") and len(set(rev.values())) == 1, rev
    assert rev["rev_default"] != digests["default"]      # (the two orders do differ in the last bits)


@pytest.mark.parametrize("case", [
    dict(G=20_000, size=96, views=1, color_sh_degree=None, feature_channels=4),                      # 4 payload channels
    dict(G=12_000, size=(70, 100), views=3, color_sh_degree=2, feature_channels=4),                  # 7 channels, ragged image
    dict(G=3_000, size=64, views=2, color_sh_degree=1, feature_channels=4, sigma_px=(4.0, 30.0), opacity_scale=1.0),   # long lists, early stops
    dict(G=300_000, size=256, views=1, color_sh_degree=None, feature_channels=4),                    # the latency workload itself
])
def test_subblock_items_forward_matches_the_serial_kernels(hip_device, case):
    """The sub-block items of k_render_fwd_small (one wave per 4x4 sub-block, four list entries per step across the wave's 16-lane rows; chosen for
    single-view-sized calls) runs the transmittance recurrence in exactly the serial kernels' order: mask (1 - T) and the
    per-pixel list prefix the backward walks must be BITWISE those of the half-tile kernel; colour, features and depth are
    per-row partial sums added once per item and may differ in the last bits only."""
    from latentsplat_amd import _lib
    case = dict(case)
    size = case.pop("size")
    H, W = (size, size) if isinstance(size, int) else size
    sc = util.make_scene(case.pop("G"), image_size=max(H, W), **case)
    bi = util.boundary_inputs(sc, H, W, bg=(0.2, 0.4, 0.6))
    runs = {}
    try:
        for name, quad, rows in (("half_tile", 0, 0), ("rows", 0, 1), ("quad", 1, 0)):
            _lib.set_knob("LSR_FWD_QUAD", quad)
            _lib.set_knob("LSR_FWD_ROWS", rows)
            r = util.HipRun(bi, hip_device)
            runs[name] = dict(mask=r.mask_out.clone(), n=torch.as_tensor(r.n_contrib().astype("int64")), T=torch.as_tensor(r.final_T()), depth=r.depth_out.clone(),
                              color=None if r.color_out is None else r.color_out.clone(), feat=None if r.feat_out is None else r.feat_out.clone())
    finally:
        _lib.set_knob("LSR_FWD_QUAD", -1)
        _lib.set_knob("LSR_FWD_ROWS", -1)
    a, q = runs["half_tile"], runs["quad"]
    assert torch.equal(a["mask"], runs["rows"]["mask"]) and torch.equal(a["depth"], runs["rows"]["depth"])
    assert torch.equal(a["mask"], q["mask"]), "mask = 1 - T must be bitwise identical"
    assert torch.equal(a["n"], q["n"]) and torch.equal(a["T"], q["T"]), "the per-pixel list prefix and final T (what the backward reads) must be identical"
    for k in ("depth", "color", "feat"):
        if a[k] is None:
            continue
        scale = max(1.0, float(a[k].abs().max()))
        assert float((a[k] - q[k]).abs().max()) <= 2e-6 * scale, k


def test_early_pair_count_equals_the_device_header(hip_device):
    """The synchronous forward's host reads the pair count from mapped host words the scanning workgroup (the last
    one of k_preprocess, or k_tile_scan for calls with more than 4096 (view, tile) counts: the 1040-pixel shape
    below) writes BEFORE it finishes the offsets; the authoritative numbers are the ones the scan leaves in the
    workspace header.  They must agree on every call — a smaller host count would under-size the binning
    workspace — and a stale sequence word must never be mistaken for this call's.  Repeated over shapes that use the
    LDS-privatised and the global tile histograms."""
    import ctypes as C
    from latentsplat_amd import _lib
    lib = _lib.load()
    for G, V, size, reps in ((120_000, 8, 256, 40), (3_000, 1, 1040, 10), (40_000, 3, 64, 40)):
        sc = util.make_scene(G, image_size=size, views=V, color_sh_degree=None, feature_channels=4, seed=3)
        bi = util.boundary_inputs(sc, size, size)          # per-view feature tensors (V,G,C), as HipRun's strides expect
        run = util.HipRun(bi, hip_device, shared_means=True)
        p = lambda x: C.c_void_p(x.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(hip_device).cuda_stream)
        for _ in range(reps):
            npairs, maxtile = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.lsr_forward_prepare(C.byref(run.d), C.byref(run.inp), p(run.geom), p(run.radii),
                                               C.byref(npairs), C.byref(maxtile), stream), "prepare")
            dp, dm, ov = C.c_int64(0), C.c_int32(0), C.c_int32(0)
            _lib.check(lib.lsr_forward_status(C.byref(run.d), p(run.geom), C.byref(dp), C.byref(dm), C.byref(ov), stream), "status")
            assert (npairs.value, maxtile.value) == (dp.value, dm.value) == (run.P, run.maxtile)


def test_speculative_forward_equals_exact_and_recovers_from_overflow(hip_device):
    """Round 5: calls of a shape that has been rendered before launch the whole forward with a workspace sized from the
    previous calls' pair counts and read the true counts afterwards (lsr_forward_speculative: no idle device between
    the host's read and the next launches).  The results must be bit for bit those of the exact (prepare + render) path —
    forward outputs, the workspaces the backward reads, gradients up to the order of their atomic sums — and a scene of
    the same shape that needs MORE pairs or LONGER lists than provided must come out right as well (detected from the
    counts, run again through the exact path)."""
    from latentsplat_amd import rasterizer as rz
    dev = hip_device
    G, V, size = 24000, 3, 96
    small = util.make_scene(G, image_size=size, views=V, color_sh_degree=1, feature_channels=4, seed=21)
    big = util.make_scene(G, image_size=size, views=V, color_sh_degree=1, feature_channels=4, seed=22, sigma_px=(2.0, 12.0))   # ~6 x the pairs
    def tensors(sc):
        bi = util.boundary_inputs(sc, size, size, bg=(0.2, 0.1, 0.0))
        return util.view_table(bi, dev), {k: bi[k].to(dev) for k in ("means", "cov6", "opac", "shs", "features")}
    gen = torch.Generator().manual_seed(3)
    g_col, g_feat = torch.randn((V, 3, size, size), generator=gen).to(dev), torch.randn((V, 4, size, size), generator=gen).to(dev)

    def run(views, t):
        leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
        out = rz.rasterize_views(views, size, size, 1, leaves["means"], leaves["cov6"], leaves["opac"], shs=leaves["shs"], features=leaves["features"])
        st = rz.last_forward_status()
        torch.autograd.backward([out[0], out[1]], [g_col, g_feat])
        return [o.detach().clone() for o in out], {k: v.grad.clone() for k, v in leaves.items()}, st

    def same(a, b, what):
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y), what + ": forward outputs"
        assert a[2] == b[2], (what, a[2], b[2])
        for k in a[1]:
            scale = max(1.0, float(b[1][k].abs().max()))
            assert float((a[1][k] - b[1][k]).abs().max()) <= 2e-5 * scale, (what, k)

    vs, ts = tensors(small)
    vb, tb = tensors(big)
    rz._ESTIMATES.clear()
    stats0 = dict(rz.SPECULATION_STATS)
    exact_small = run(vs, ts)                      # first call of the shape: exact path, records the estimate
    assert rz.SPECULATION_STATS["exact"] == stats0["exact"] + 1 and rz.SPECULATION_STATS["speculative"] == stats0["speculative"]
    spec_small = run(vs, ts)                       # second call: speculative, fits
    assert rz.SPECULATION_STATS["speculative"] == stats0["speculative"] + 1
    same(spec_small, exact_small, "speculative == exact")
    spec_big = run(vb, tb)                         # same shape, far more pairs and longer lists: overflow -> exact re-run
    assert rz.SPECULATION_STATS["reruns"] == stats0["reruns"] + 1
    assert spec_big[2]["num_pairs"] > 2 * exact_small[2]["num_pairs"]
    again_big = run(vb, tb)                        # the estimate has grown: speculative again
    assert rz.SPECULATION_STATS["speculative"] == stats0["speculative"] + 2
    rz._ESTIMATES.clear()
    exact_big = run(vb, tb)
    same(spec_big, exact_big, "overflow re-run == exact")
    same(again_big, exact_big, "speculative after the re-run == exact")
    # a hint that is too small on its own (the pair capacity fits): lists beyond it must be caught as well
    key = next(iter(rz._ESTIMATES))
    rz._ESTIMATES[key] = (rz._ESTIMATES[key][0], 16)
    reruns = rz.SPECULATION_STATS["reruns"]
    low_hint = run(vb, tb)
    assert rz.SPECULATION_STATS["reruns"] == reruns + 1
    same(low_hint, exact_big, "too small a tile hint == exact")


def test_front_half_launched_early_equals_the_one_call_forward(hip_device):
    """ABI v10: the autograd op launches the front half (projection, key emission, tile scan) through lsr_forward_front as soon as
    geom_ws and radii exist and allocates the rest while the device works; the full call follows with LSR_FWD_FRONT_DONE.  The
    launches and their stream order are those of the one-call form: every output bit for bit, on the exact, the speculative and
    the no-sync path; a full call whose pending front half does not match is refused (LSR_EINVAL) without launching anything."""
    import ctypes as C
    from latentsplat_amd import _lib
    from latentsplat_amd import rasterizer as rz
    from latentsplat_amd._lib import Dims, Outputs
    dev = hip_device
    G, V, size = 20_000, 2, 96
    sc = util.make_scene(G, image_size=size, views=V, color_sh_degree=1, feature_channels=4, seed=17)
    bi = util.boundary_inputs(sc, size, size, bg=(0.1, 0.2, 0.3))
    views = util.view_table(bi, dev)
    t = {k: bi[k].to(dev) for k in ("means", "cov6", "opac", "shs", "features")}
    call = lambda **kw: rz.rasterize_views(views, size, size, 1, t["means"], t["cov6"], t["opac"], shs=t["shs"], features=t["features"], **kw)
    got = {}
    saved = rz._EARLY_FRONT
    try:
        for early in (True, False):
            rz._EARLY_FRONT = early
            rz._ESTIMATES.clear()
            stats = dict(rz.SPECULATION_STATS)
            with torch.no_grad():
                a = call()                          # first call of the shape: exact path (nothing to launch early)
                b = call()                          # speculative
                assert rz.SPECULATION_STATS["speculative"] == stats["speculative"] + 1
                st = rz.last_forward_status()
                c = call(pair_capacity=int(1.3 * st["num_pairs"]) + 64, max_tile_hint=st["max_tile_pairs"])
                assert not rz.last_forward_status()["overflow"]
            got[early] = [[o.clone() for o in x] for x in (a, b, c)]
    finally:
        rz._EARLY_FRONT = saved
    ref = got[False][0]
    for early in (True, False):
        for which, outs in zip(("exact", "speculative", "no-sync"), got[early]):
            for x, y in zip(outs, ref):
                assert torch.equal(x, y), (early, which)
    # gradients through the early-front speculative path (the workspaces the backward reads are the one-call form's)
    grads = {}
    try:
        for early in (True, False):
            rz._EARLY_FRONT = early
            leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
            out = rz.rasterize_views(views, size, size, 1, leaves["means"], leaves["cov6"], leaves["opac"], shs=leaves["shs"], features=leaves["features"])
            gen = torch.Generator().manual_seed(5)
            torch.autograd.backward([out[0], out[1]], [torch.randn(out[0].shape, generator=gen).to(dev), torch.randn(out[1].shape, generator=gen).to(dev)])
            grads[early] = {k: v.grad.clone() for k, v in leaves.items()}
    finally:
        rz._EARLY_FRONT = saved
    for k in grads[True]:
        scale = max(1.0, float(grads[False][k].abs().max()))
        assert float((grads[True][k] - grads[False][k]).abs().max()) <= 2e-5 * scale, k

    # the handshake at the C ABI
    lib = _lib.load()
    run = util.HipRun(bi, dev)
    p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cap = run.P + 100
    d = Dims.from_buffer_copy(run.d)
    d.forward_flags |= _lib.FWD_FRONT_DONE
    binws = torch.zeros(lib.lsr_binning_workspace_bytes(C.byref(d), cap, 2 ** 31 - 1), dtype=torch.uint8, device=dev)
    outs = Outputs(p(run.color_out), p(run.feat_out), p(run.mask_out), p(run.depth_out), p(run.radii))
    nosync = lambda capacity: lib.lsr_forward_nosync(C.byref(d), C.byref(run.inp), p(run.geom), p(binws), p(run.img), capacity, run.maxtile, C.byref(outs), stream)
    EINVAL = -1
    assert lib.lsr_error_string(EINVAL) and nosync(cap) == EINVAL, "the flag without a pending front half"
    want = [x.clone() for x in (run.color_out, run.feat_out, run.mask_out, run.depth_out)]
    for x in (run.color_out, run.feat_out, run.mask_out, run.depth_out):
        x.fill_(-7.0)
    _lib.check(lib.lsr_forward_front(C.byref(d), C.byref(run.inp), p(run.geom), p(run.radii), cap, stream), "front")
    assert nosync(cap + 1) == EINVAL, "another capacity than the front half's"
    torch.cuda.synchronize(dev)
    assert float(run.mask_out.max()) == -7.0, "a refused call must not launch anything"
    assert nosync(cap) == EINVAL, "the pending front half is consumed by the refused call"
    _lib.check(lib.lsr_forward_front(C.byref(d), C.byref(run.inp), p(run.geom), p(run.radii), cap, stream), "front")
    _lib.check(nosync(cap), "no-sync after front")
    torch.cuda.synchronize(dev)
    for x, y in zip((run.color_out, run.feat_out, run.mask_out, run.depth_out), want):
        assert torch.equal(x, y)


@pytest.mark.parametrize("rows", [0, 1])
@pytest.mark.parametrize("record", [0, 1])
def test_finished_subblocks_leave_the_compaction_without_changing_anything(hip_device, rows, record):
    """Round 6: once every pixel of a 4x4 sub-block has run out of transmittance the forward kernels stop listing entries for it
    (its lanes would blend nothing; its list would still set the lock-step length of every batch).  LSR_FWD_LIVE=0 keeps the
    round-5 behaviour: images, final T, n_contrib and — in a forward a backward follows — the narrowed render lists must be the
    same bit for bit, for the half-tile kernel and the row items, on a scene whose pixels do finish (and a ragged image edge)."""
    from latentsplat_amd import _lib
    sc = util.make_scene(5_000, image_size=72, views=3, color_sh_degree=1, feature_channels=4, sigma_px=(2.0, 14.0), opacity_scale=1.0, seed=29)
    bi = util.boundary_inputs(sc, 72, 60, bg=(0.3, 0.1, 0.2))
    runs = {}
    try:
        _lib.set_knob("LSR_FWD_ROWS", rows)
        _lib.set_knob("LSR_FWD_QUAD", 0)
        for live in (1, 0):
            _lib.set_knob("LSR_FWD_LIVE", live)
            runs[live] = util.HipRun(bi, hip_device, forward_flags=_lib.FWD_FOR_BACKWARD if record else 0)
    finally:
        _lib.set_knob("LSR_FWD_LIVE", 1)
        _lib.set_knob("LSR_FWD_ROWS", -1)
        _lib.set_knob("LSR_FWD_QUAD", -1)
    a, b = runs[1], runs[0]
    assert float((a.mask_out > 1.0 - 2e-4).float().mean()) > 0.2, "the scene must saturate a good part of its pixels"
    for x, y in ((a.color_out, b.color_out), (a.feat_out, b.feat_out), (a.mask_out, b.mask_out), (a.depth_out, b.depth_out)):
        assert torch.equal(x, y)
    assert np.array_equal(a.n_contrib(), b.n_contrib()) and np.array_equal(a.final_T(), b.final_T())
    assert np.array_equal(a.half_count(), b.half_count())
    assert np.array_equal(a.half_list(), b.half_list()), "the narrowed lists must not depend on the compaction's masks"
    # (an entry staged for no live sub-block is not looked at any more: the "steep" item flags may only shrink)
    (fa, va), (fb, vb) = a.item_flags(), b.item_flags()
    assert va == vb and not (fa & ~fb).any()
