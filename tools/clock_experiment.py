#!/usr/bin/env python
"""why does k_render_fwd take ~0.235 ms in the forward-only loop and ~0.200 ms inside fwd+bwd
steps of the same run?  Measures the kernel (hipEvents inside the library) and the SHADER CLOCK (a one-wave probe
kernel on a second stream: core-clock cycles per constant-rate wall-clock tick, tools/microbench/clock_probe.hip)
in several loops over the bench workload:
   fwd_tight      : back-to-back forward steps, only render_forward bracketed (the headline's timed region)
   fwd_all_events : the same with every stage bracketed
   fwd_idle_1ms   : a device-idle gap of ~1 ms after every step (host sleeps after a synchronise)
   fwd_idle_5ms
   fwdbwd         : forward + backward steps (every stage bracketed, as bench.py measures kernel_ms_fwdbwd)
   fwd_then_clear : forward steps, each followed by a 307 MB device memset (what the backward's workspace clear does)
Prints one JSON line per mode; `rocm-smi` clocks / power are sampled once in the middle of each mode."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd import _lib  # noqa: E402
from latentsplat_amd.rasterizer import rasterize_views  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        card = next(iter(j.values()))
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}


def main():
    dev = torch.device("cuda", 0)
    _lib.load()
    probe = C.CDLL(os.path.join(ROOT, "tools", "microbench", "libclock_probe.so"))
    probe.clock_probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    khz = probe.clock_probe_wall_khz()
    inp = bench.build_inputs(300_000, 16, 256, dev, 1234)
    gf = torch.randn((16, 4, 256, 256), device=dev)
    big = torch.empty(307_200_000, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)
    slots = 4096
    buf = torch.zeros(2 * slots, dtype=torch.int64, device=dev)

    def fwd():
        with torch.no_grad():
            rasterize_views(inp["views"], 256, 256, 0, inp["means"], inp["cov"], inp["opac"], features=inp["features"])

    def fb():
        m, c, o, f = (t.detach().requires_grad_(True) for t in (inp["means"], inp["cov"], inp["opac"], inp["features"]))
        rasterize_views(inp["views"], 256, 256, 0, m, c, o, features=f)[1].backward(gf)

    def run(mode, seconds=2.5):
        step, only, gap, clear = fwd, ("render_forward",), 0.0, False
        if mode == "fwd_all_events":
            only = ()
        elif mode == "fwd_idle_1ms":
            gap = 1e-3
        elif mode == "fwd_idle_5ms":
            gap = 5e-3
        elif mode == "fwdbwd":
            step, only = fb, ()
        elif mode == "fwd_then_clear":
            clear = True
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        buf.zero_()
        _lib.profile_read()
        if only:
            _lib.profile_enable(True, only=only)
        else:
            _lib.profile_enable(True)
        box = {}
        th = threading.Timer(seconds / 2, lambda: box.update(smi=smi()))
        th.start()
        n, slot, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            step()
            if clear:
                big.zero_()
            if n % 8 == 0 and slot < slots:
                probe.clock_probe_launch(C.c_void_p(buf.data_ptr()), slot, 200, C.c_void_p(side.cuda_stream))
                slot += 1
            if gap:
                torch.cuda.synchronize(dev)
                time.sleep(gap)
            n += 1
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        _lib.profile_enable(False)
        prof = _lib.profile_read()
        th.join()
        b = buf.cpu().numpy().reshape(-1, 2)[:slot]
        b = b[b[:, 1] > 0]
        mhz = b[:, 0] / b[:, 1] * (khz / 1e3)
        half = len(mhz) // 2
        res = {"mode": mode, "steps": n, "ms_per_step_wall": round(1e3 * el / n, 4),
               "kernel_ms": {k: round(ms / c, 4) for k, (ms, c) in prof.items() if c},
               "sclk_mhz": {"n": int(len(mhz)), "mean": round(float(mhz.mean()), 1) if len(mhz) else None,
                            "first_half_mean": round(float(mhz[:half].mean()), 1) if half else None,
                            "second_half_mean": round(float(mhz[half:].mean()), 1) if half else None,
                            "min": round(float(mhz.min()), 1) if len(mhz) else None, "max": round(float(mhz.max()), 1) if len(mhz) else None},
               "wall_clock_khz": khz, "rocm_smi": box.get("smi")}
        print(json.dumps(res), flush=True)

    def cold_start(idle_s, steps=23):
        """What bench.py's headline region looked like in rounds 1-3: an idle device, then 3 + 20 forward steps (10 ms)."""
        torch.cuda.synchronize(dev)
        time.sleep(idle_s)
        buf.zero_()
        _lib.profile_read()
        _lib.profile_enable(True, only=("render_forward",))
        t0 = time.perf_counter()
        for i in range(steps):
            fwd()
            probe.clock_probe_launch(C.c_void_p(buf.data_ptr()), i, 100, C.c_void_p(side.cuda_stream))
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        _lib.profile_enable(False)
        prof = _lib.profile_read()
        b = buf.cpu().numpy().reshape(-1, 2)[:steps]
        mhz = [round(float(c / w * khz / 1e3), 0) if w > 0 else None for c, w in b]
        print(json.dumps({"mode": f"cold_start_after_{idle_s}s_idle", "steps": steps, "ms_per_step_wall": round(1e3 * el / steps, 4),
                          "kernel_ms": {k: round(ms / c, 4) for k, (ms, c) in prof.items() if c}, "sclk_mhz_per_step": mhz}), flush=True)

    if "--cold-only" not in sys.argv:
        for mode in ("fwd_tight", "fwdbwd", "fwd_all_events", "fwd_idle_1ms", "fwd_idle_5ms", "fwd_then_clear", "fwd_tight", "fwdbwd"):
            run(mode)
    for idle in (2.0, 0.5, 2.0):
        cold_start(idle)
    run("fwd_tight", 1.0)
    run("fwdbwd", 1.0)


if __name__ == "__main__":
    main()
