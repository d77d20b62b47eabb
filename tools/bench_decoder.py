#!/usr/bin/env python
"""End-to-end timing of the decoder surface (what the reference's training loop calls):
DecoderSplattingCUDA.forward (+ backward) at the reference's experiment shapes (BASELINE configs[3]/[4]:
G = 393 216 Gaussians per scene, colour SH degree 4 + 4-channel latent SH degree 2, 256x256).
Splits host-side torch work from rasterizer-kernel time (lsr_profile hook)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentsplat_amd import _lib, decoder as dec  # noqa: E402
from latentsplat_amd.synthetic import make_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=1)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--gaussians", type=int, default=393_216)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    b, v, G = args.scenes, args.views, args.gaussians
    scenes = [make_scene(G, image_size=256, views=v, color_sh_degree=4, feature_channels=4, feature_sh_degree=2,
                         seed=1234 + s).to(dev) for s in range(b)]
    st = lambda n: torch.stack([getattr(s, n) for s in scenes])
    gauss = dec.Gaussians(st("means").requires_grad_(True), st("covariances").requires_grad_(True),
                          st("opacities").requires_grad_(True), st("color_sh").requires_grad_(True),
                          st("feature_sh").requires_grad_(True))
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.0, 0.0, 0.0]).to(dev)
    ext, intr, near, far = st("extrinsics"), st("intrinsics"), st("near"), st("far")

    def fwd():
        return d.forward(gauss, ext, intr, near, far, (256, 256))

    def fwdbwd():
        out = fwd()
        loss = out.color.square().mean() + out.feature_posterior.mean.square().mean()
        loss.backward()
        for t in (gauss.means, gauss.covariances, gauss.opacities, gauss.color_harmonics, gauss.feature_harmonics):
            t.grad = None

    res = {}
    for name, fn in (("forward", lambda: fwd()), ("forward_backward", fwdbwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_read()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        prof = _lib.profile_read(); _lib.profile_enable(False)
        kern = sum(ms for ms, n in prof.values()) / args.steps
        res[name] = dict(ms_per_step=1e3 * el, views_per_s=b * v / el, rasterizer_kernel_ms=kern,
                         host_and_torch_ms=1e3 * el - kern,
                         kernels={k: ms / args.steps for k, (ms, n) in prof.items() if n})
    print(json.dumps(dict(scenes=b, views=v, gaussians=G, **res)))


if __name__ == "__main__":
    main()
