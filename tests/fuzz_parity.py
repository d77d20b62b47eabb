#!/usr/bin/env python
"""Randomised parity sweep (test infrastructure, like everything under tests/): random scene / image / payload shapes
through exactly the checks of tests/test_parity_gpu.py — forward: radii, rectangles, depth bits, pixel means, conics, tile
offsets and sorted lists bit for bit against the CPU oracle, images at 1e-4; backward: every input gradient at the
suite's bars.  The shapes the suite pins by hand are a dozen; this draws them from the corners the kernels switch on
(Gaussian counts around the 64-wide blocks, image sides around the 16-pixel tiles, 1..6 views, colour off / SH degree
0..4, 0..32 feature channels with latent SH degree 0..2, tiny and huge splats, both projection conventions; a fifth of
the draws go through the scene-level inputs of the decoder path — in-kernel scene scale, 3x3 covariances, stored-layout
colour SH, latent SH evaluated in-kernel, shared by all views (the fused projection + SH kernel) or per view; an eighth
compare the no-sync (latency) forward bit for bit with the synchronous one).

    python -m tests.fuzz_parity --seconds 240 --seed 1 [--out gpurun_out/fuzz_parity.json]

The suite's BUDGETS for how much of a scene may sit next to a discontinuity (2 % of the pixels, 1 % of the Gaussians: tuned
to its own scenes) are lifted here — a 7 x 50 image under 30-pixel splats has more — while every bar stays: pixels and
gradient rows without a fragile evaluation nearby are held to 1e-4 (images: abs; gradients: of the tensor's scale), fragile
ones to their flip bounds.  The suite additionally holds CLEAN gradient rows of its own scenes to 2e-5 of scale; here the
worst clean row of every case is collected instead (`worst_clean_row_over_scale`, `cases_over_suite_clean_bar`).
Failures are classed `exact` (an integer / bit-pattern comparison), `bar` (a float tolerance) or `other`, and recorded with
the case that reproduces them (`--only N` re-runs draw N of a seed)."""
import argparse
import json
import random
import time
import traceback

import torch

from tests import test_parity_gpu as tp
from tests import util

G_CHOICES = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 1000, 2047, 2048, 4097, 6000, 12000]
SIDE_CHOICES = [1, 7, 15, 16, 17, 31, 32, 33, 48, 50, 64, 70, 96, 100, 130]


def draw(rng: random.Random) -> dict:
    color = rng.choice([None, 0, 1, 2, 3, 4])
    fc = rng.choice([None, 1, 3, 4, 5, 8, 9, 12, 16, 32])
    if color is None and fc is None:
        fc = 4
    case = dict(G=rng.choice(G_CHOICES), size=(rng.choice(SIDE_CHOICES), rng.choice(SIDE_CHOICES)), views=rng.choice([1, 1, 2, 3, 4, 5]),
                color_sh_degree=color, feature_channels=fc, sigma_px=rng.choice([(0.3, 3.0), (0.05, 0.5), (2.0, 20.0), (4.0, 30.0)]),
                opacity_scale=rng.choice([1.0 / 3.0, 1.0, 0.05]), seed=rng.randrange(1 << 30))
    if fc:
        case["feature_sh_degree"] = rng.choice([0, 0, 1, 2])
    return case


class lifted_budgets:
    """Context: the suite's fragile-fraction budgets lifted, its clean-row bar at 1e-4 (restored on exit: the functions are
    module attributes the rest of the suite uses too)."""

    def __enter__(self):
        import functools
        self.saved = (util.assert_close_except_fragile, util.assert_grad_close_except_fragile, tp.CLEAN_TOL)
        util.assert_close_except_fragile = functools.partial(self.saved[0], max_fragile_frac=1.0)
        util.assert_grad_close_except_fragile = functools.partial(self.saved[1], max_direct_frac=1.0, min_strict=0.0)
        tp.CLEAN_TOL = tp.ABS_TOL
        return self

    def __exit__(self, *exc):
        util.assert_close_except_fragile, util.assert_grad_close_except_fragile, tp.CLEAN_TOL = self.saved
        return False


def classify(msg: str) -> str:
    if "n_considered mismatch fraction" in msg:   # every mismatching pixel was a fragile one (the assertion before it): a budget, lifted
        return "budget"
    exact = ("radii", "tile rectangles", "depth bits", "pixel means", "conic", "tile counts", "tile starts", "sorted tile lists", "n_considered", "run.P")
    if any(k in msg for k in exact) or "Arrays are not equal" in msg or "no-sync forward differs" in msg or "last_forward_status" in msg:
        return "exact"
    if any(k in msg for k in ("off by", "off the", "moved more", "next to a fragile", "within the")):
        return "bar"
    return "other"


def run_case(dev, case: dict, mode: str, contracted: bool):
    from latentsplat_amd import _lib
    from oracle import oracle as orc
    lib = _lib.load()
    try:
        if contracted:
            lib.lsr_set_projection_contraction(1)
            orc.set_fma_contraction(True)
        if mode.startswith("fused"):   # scene-level inputs: in-kernel scene scale, 3x3 covariances, stored-layout colour SH, latent SH
            cfg = dict(case, size=max(case["size"]), feature_channels=case["feature_channels"] or 4, views=max(1, case["views"] + (mode == "fused_shared")))
            cfg.setdefault("feature_sh_degree", 2)
            while cfg["feature_channels"] * (cfg["feature_sh_degree"] + 1) ** 2 > 120:   # the fused latent-SH contract (lsr_rasterizer.h); beyond it the
                cfg["feature_sh_degree"] -= 1                                              # decoder evaluates the harmonics in torch, like the reference
            tp.test_fused_scene_inputs_match_oracle(dev, cfg, mode == "fused_shared")
        elif mode == "nosync":   # the latency path (device-side pair count, later sort tiers in one launch) against the synchronous forward
            from latentsplat_amd.rasterizer import last_forward_status, rasterize_views
            sc, H, W = tp._scene(case)
            bi = util.boundary_inputs(sc, H, W, bg=(0.1, 0.2, 0.3))
            t = {k: (None if bi[k] is None else bi[k].to(dev)) for k in ("means", "cov6", "opac", "shs", "features")}
            call = lambda **kw: rasterize_views(util.view_table(bi, dev), H, W, bi["sh_degree"], t["means"], t["cov6"], t["opac"], shs=t["shs"], features=t["features"], **kw)
            with torch.no_grad():
                ref = call()
                st = last_forward_status()
                hint = random.Random(case["seed"]).choice([64, 1024, max(1, st["max_tile_pairs"]), 16384])
                out = call(pair_capacity=max(1, int(1.25 * st["num_pairs"]) + 1), max_tile_hint=hint)
                assert last_forward_status() == st, (last_forward_status(), st)
            for a, b in zip(out, ref):
                assert (a is None) == (b is None)
                assert a is None or torch.equal(a, b), "no-sync forward differs from the synchronous forward"
        elif mode == "forward":
            sc, H, W = tp._scene(case)
            bi = util.boundary_inputs(sc, H, W, bg=(0.2, 0.4, 0.6))
            run = util.HipRun(bi, dev)
            tp._check_forward(bi, run)
            if case["seed"] % 3 == 0 and not contracted:     # round 6: the reached-only binning against the published lists it thins out
                from latentsplat_amd import _lib as _l
                from tests.test_reached_only_gpu import compare_runs
                compare_runs(run, util.HipRun(bi, dev, forward_flags=_l.FWD_REACHED_ONLY))
        else:
            tp._grad_case(dev, case, mode == "backward_aux")
    finally:
        if contracted:
            lib.lsr_set_projection_contraction(0)
            orc.set_fma_contraction(False)


def sweep(dev, seed: int, seconds: float, max_cases: int = 0, only: int = -1) -> dict:
    rng = random.Random(seed)
    t0 = time.time()
    done, failures, by_mode = 0, [], {}
    worst_clean, over_suite_bar = 0.0, 0
    n = 0
    with lifted_budgets():
        while (time.time() - t0 < seconds and (not max_cases or done < max_cases)) or (only >= 0 and n <= only):
            case = draw(rng)
            mode = rng.choice(["forward", "forward", "forward", "backward", "backward_aux", "fused_shared", "fused_per_view", "nosync"])
            contracted = mode == "forward" and rng.random() < 0.25
            if mode != "forward":   # the gradient oracle is the slow part: keep its cases small
                case["G"] = min(case["G"], 4097)
            n += 1
            if only >= 0 and n - 1 != only:
                continue
            n0 = len(util.ACCOUNTING)
            try:
                run_case(dev, dict(case), mode, contracted)
                by_mode[mode] = by_mode.get(mode, 0) + 1
            except Exception as e:   # noqa: BLE001 — every failure is a finding to record, whatever its type
                msg = f"{type(e).__name__}: {str(e)[:400]}"
                failures.append(dict(draw=n - 1, mode=mode, contracted=contracted, kind=classify(msg), case=case, error=msg))
                traceback.print_exc()
            done += 1
            clean = [a["worst_strict_err"] / a["scale"] for a in util.ACCOUNTING[n0:] if a["kind"] == "grad"]
            del util.ACCOUNTING[n0:]   # the sweep's comparisons stay out of the suite's parity accounting
            if clean:
                worst_clean = max(worst_clean, max(clean))
                over_suite_bar += max(clean) > 2e-5
            if only >= 0:
                break
    failures = [f for f in failures if f["kind"] != "budget"]
    kinds = {k: sum(f["kind"] == k for f in failures) for k in ("exact", "bar", "other")}
    return dict(seed=seed, seconds=round(time.time() - t0, 1), cases=done, passed=done - len(failures), passed_by_mode=by_mode,
                failed_by_kind=kinds, worst_clean_row_over_scale=worst_clean, cases_over_suite_clean_bar=int(over_suite_bar), failures=failures)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run only draw N of the seed")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    res = sweep(torch.device("cuda", 0), args.seed, args.seconds, only=args.only)
    print(json.dumps(res, indent=1, default=str))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1, default=str)
    return 1 if res["failures"] else 0


if __name__ == "__main__":
    raise SystemExit(main())
