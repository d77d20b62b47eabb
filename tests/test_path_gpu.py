"""The hot path CHAINED the way the reference's training step runs it (/root/reference/src/model/model_wrapper.py:361-385):

    GaussianAdapter tail (raw network output -> means / covariances)          k_adapter_fwd / k_adapter_bwd
    -> DecoderSplattingCUDA.forward (preprocess, binning, SH, compositing)     the rasterizer kernels
    -> posterior sample + antialiased 1/8 rescale + skip concatenation         k_latent_fwd / k_latent_bwd

at BASELINE configs[3]'s shape — 2 context cameras x 256^2 rays x 3 depth samples = 393 216 Gaussians, colour SH
degree 4 + 4-channel latent SH degree 2, 1 x 4 target views at 256x256 — forward and backward, with the gradient
flowing from the latent / skip / colour heads back to the RAW adapter inputs (coordinates, depths, raw scales and
rotations).  Checked against the three CPU oracles chained the same way (oracle/adapter_oracle.py ->
raster_oracle.c through tests/util.to_boundary -> oracle/latent_oracle.py).  Each piece has its own parity tests;
this one pins their composition (layouts handed from kernel to kernel, gradient routing through all three
autograd nodes)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
SIZE, CAMS, RAYS, S = 256, 2, 256 * 256, 3
G = CAMS * RAYS * S


def make_path_inputs(seed=77):
    """Raw adapter inputs of two context cameras + the rest of a configs[3] scene (CPU tensors)."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(SIZE), torch.arange(SIZE), indexing="ij")
    coords = ((torch.stack([xs, ys], -1).reshape(RAYS, 2).float() + 0.5) / SIZE)[None].repeat(CAMS, 1, 1)
    E = torch.eye(4).repeat(CAMS, 1, 1)
    E[1, 0, 3] = 0.3
    K = torch.tensor([[0.8, 0, 0.5], [0, 0.8, 0.5], [0, 0, 1.0]]).repeat(CAMS, 1, 1)
    depths = 1.5 + 8.0 * torch.rand(CAMS, RAYS, S, generator=g)
    raw = torch.randn(CAMS, RAYS, 7, generator=g)
    sc = util.make_scene(G, image_size=SIZE, views=4, color_sh_degree=4, feature_channels=4, feature_sh_degree=2, seed=seed)
    return dict(E=E, K=K, coords=coords, depths=depths, raw=raw, opac=sc.opacities, csh=sc.color_sh, fsh=sc.feature_sh,
                extrinsics=sc.extrinsics, intrinsics=sc.intrinsics, near=sc.near, far=sc.far)


def hip_path(inp, dev, heads=None, noise=None):
    """adapter -> decoder -> epilogue on the MI355X; returns (leaves, decoder output, epilogue)."""
    from latentsplat_amd import decoder as dec
    from latentsplat_amd.decoder.latent_epilogue import decoder_output_epilogue
    from latentsplat_amd.gaussian_adapter import adapter_geometry
    t = lambda k: inp[k].to(dev)
    leaves = {k: t(k).clone().requires_grad_(True) for k in ("coords", "depths", "raw")}
    means, cov, _, _ = adapter_geometry(t("E"), t("K"), leaves["coords"], leaves["depths"], leaves["raw"], (SIZE, SIZE), 0.5, 15.0)
    gauss = dec.Gaussians(means.reshape(1, G, 3), cov.reshape(1, G, 3, 3), t("opac")[None], t("csh")[None], t("fsh")[None])
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda"), [0.1, 0.2, 0.3]).to(dev)
    out = d.forward(gauss, t("extrinsics")[None], t("intrinsics")[None], t("near")[None], t("far")[None], (SIZE, SIZE))
    ep = decoder_output_epilogue(out, 8, noise=noise)
    return leaves, out, ep, (means.detach().reshape(G, 3), cov.detach().reshape(G, 3, 3))


def test_adapter_decoder_epilogue_chain_against_chained_oracles(hip_device):
    from latentsplat_amd.rasterizer import build_view_table
    from oracle import adapter_oracle as ao
    from oracle import latent_oracle as lo
    dev = hip_device
    inp = make_path_inputs()
    gen = torch.Generator().manual_seed(5)
    noise = torch.randn(1, 4, 4, SIZE, SIZE, generator=gen)
    g_z = torch.randn(1, 4, 4, SIZE // 8, SIZE // 8, generator=gen)
    g_skip = torch.randn(1, 4, 7, SIZE, SIZE, generator=gen)
    g_color = torch.randn(1, 4, 3, SIZE, SIZE, generator=gen)

    leaves, out, ep, (means_hip, cov_hip) = hip_path(inp, dev, noise=noise.to(dev))
    torch.autograd.backward([ep.z, ep.skip_z, out.color], [g_z.to(dev), g_skip.to(dev), g_color.to(dev)])
    assert ep.z.shape == (1, 4, 4, SIZE // 8, SIZE // 8) and ep.skip_z.shape == (1, 4, 7, SIZE, SIZE)

    # ---- the same chain on the CPU oracles ----
    cl = {k: inp[k].clone().requires_grad_(True) for k in ("coords", "depths", "raw")}
    means_o, cov_o, _, _ = ao.adapter_geometry(inp["E"], inp["K"], cl["coords"], cl["depths"], cl["raw"][..., :3], cl["raw"][..., 3:7],
                                               (SIZE, SIZE), 0.5, 15.0)
    means_o, cov_o = means_o.reshape(G, 3), cov_o.reshape(G, 3, 3)
    # The two adapters agree to float rounding (tests/test_adapter_gpu.py) but not bit for bit, and a last-bit change of
    # a mean moves alpha of a sub-pixel splat by more than the oracle's fragile window.  The rasterizer oracle is
    # therefore evaluated on EXACTLY what the HIP rasterizer was handed (the HIP adapter's outputs, read back); its
    # gradient with respect to them is then pushed through the adapter oracle's autograd graph.
    assert float((means_hip.cpu() - means_o).abs().max()) <= 2e-5 * float(means_o.abs().max())
    assert float((cov_hip.cpu() - cov_o).abs().max()) <= 2e-5 * float(cov_o.abs().max())
    means, cov = means_hip.cpu().clone().requires_grad_(True), cov_hip.cpu().clone().requires_grad_(True)
    bg = (0.1, 0.2, 0.3)
    views_cpu = build_view_table(inp["extrinsics"].to(dev), inp["intrinsics"].to(dev), inp["near"].to(dev), inp["far"].to(dev),
                                 torch.tensor(bg, device=dev), True).cpu()
    n = lambda t: None if t is None else t.detach().contiguous().numpy()
    frag_d, frag_b = [], []
    for v in range(4):
        m, c6, op, sh, cp, ft = util.to_boundary(views_cpu, v, means, cov, inp["opac"][:, None], inp["csh"], None, None, inp["fsh"], True)
        vw = views_cpu[v]
        view = util.orc.View(SIZE, SIZE, float(vw[35]), float(vw[36]), vw[37:40].numpy(), vw[0:16].numpy().reshape(4, 4),
                             vw[16:32].numpy().reshape(4, 4), vw[32:35].numpy(), 4)
        o = util.orc.forward(view, n(m), n(c6), n(op), n(sh), None, n(ft))
        d_, b_ = util.fragile_gaussians(o, SIZE)
        frag_d.append(d_); frag_b.append(b_)
        # epilogue on the oracle's images (torch autograd), its gradient back into the rasterizer oracle
        feat = torch.from_numpy(o["feature"]).requires_grad_(True)
        e = lo.latent_epilogue(feat, torch.from_numpy(o["mask"]), noise[0, v], torch.from_numpy(o["color"]), 8)
        util.assert_close_except_fragile(ep.skip_z[0, v].detach().cpu().numpy(), e["skip"].detach().numpy(), o, 1e-4, f"path skip[view {v}]",
                                         max_fragile_frac=0.08)   # pixel-aligned sub-pixel splats: many evaluations sit near the alpha threshold
        zerr = (ep.z[0, v].detach().cpu() - e["z"].detach()).abs()
        assert float((zerr > 1e-4).float().mean()) <= 5e-3 and float(zerr.max()) <= 1e-3, f"path z[view {v}]: max err {float(zerr.max()):.2e}"
        torch.autograd.backward([e["z"], e["skip"]], [g_z[0, v], g_skip[0, v]])
        b = util.orc.backward(view, n(m), n(c6), n(op), n(sh), None, n(ft), o, g_color[0, v].numpy(), feat.grad.numpy())
        # (colour SH coefficients are not leaves here; the direction term of their evaluation is part of b["means3D"])
        torch.autograd.backward([m, c6, ft], [torch.from_numpy(np.ascontiguousarray(b[k])) for k in ("means3D", "cov3D", "features")])
    torch.autograd.backward([means_o, cov_o], [means.grad, cov.grad])
    direct = np.unique(np.concatenate(frag_d))
    behind = np.setdiff1d(np.unique(np.concatenate(frag_b)), direct)
    # Gaussian (cam, ray, sample) -> row of the per-ray tensors (cam * RAYS + ray) / element of depths
    rows = lambda idx: np.unique(idx // S)
    for k, got, want, dmap in (("coords", leaves["coords"].grad, cl["coords"].grad, rows), ("raw", leaves["raw"].grad, cl["raw"].grad, rows),
                               ("depths", leaves["depths"].grad, cl["depths"].grad, lambda idx: idx)):
        got = got.cpu().numpy().reshape(-1, 1) if k == "depths" else got.cpu().numpy().reshape(CAMS * RAYS, -1)
        want = want.numpy().reshape(got.shape)
        dd, bb = dmap(direct), np.setdiff1d(dmap(behind), dmap(direct))
        util.assert_grad_close_except_fragile(got, want, dd, bb, 1e-4, f"path dL/d{k}", clean_tol=5e-5, max_direct_frac=0.1, min_strict=0.85)
