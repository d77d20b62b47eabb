#!/usr/bin/env python
"""Latency of calls into an idle device with the front half launched early (ABI v10 lsr_forward_front) or not: bench.py's
`latency` legs (synchronised / streamed calls of 1 and 4 views, the reference's per-view call pattern), alternating.

    python tools/ab_early_front.py [--rounds 3]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from latentsplat_amd import rasterizer as rz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for r in range(args.rounds):
        for early in (True, False):
            rz._EARLY_FRONT = early
            lat = bench.latency_timing(dev, 300_000, 256, 1234, iters=args.iters)
            row = dict(round=r, early_front=early)
            for k in ("views_1", "views_4"):
                for mode in ("sync", "nosync"):
                    row[f"{k}_{mode}_synced_ms"] = round(lat[k][mode]["ms_per_call_synced"], 4)
                    row[f"{k}_{mode}_streamed_ms"] = round(lat[k][mode]["ms_per_call_streamed"], 4)
            row["per_view_loop_ms"] = round(1e3 * lat["dropin_per_view_loop"]["seconds_per_view_device_complete"], 4)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
