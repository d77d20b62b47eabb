"""Differentiable PyTorch-CPU restatement of the rasterizer (ORACLE — test infrastructure only,
parity unpinned; see raster_oracle.c header for what it follows).

Independent of raster_oracle.c's hand-derived backward: gradients here come from autograd, so
agreement between the two pins the analytic backward:
  (a) straight-through ``min(0.99, .)``; (b) skipped contributions are masked, not branched;
  (c) early termination is a stop mask computed under ``no_grad``; (d) the 1.3*tanfov clamp
  zeroes d/dt.x, d/dt.y and treats the clamped t.x as constant w.r.t. t.z.

Dense (pixels x Gaussians) evaluation: small cases only (G <= ~2k, <= 64x64).
"""
from __future__ import annotations

import torch

from . import oracle as _o  # noqa: F401  (same package; constants shared by value below)

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
SH_C4 = [2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892,
         0.10578554691520431, -0.6690465435572892, 0.47308734787878004, -1.7701307697799304,
         0.6258357354491761]


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    """(G,3) unit dirs -> (G,(deg+1)^2) basis in the upstream axis convention."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if deg >= 1:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg >= 3:
        b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    if deg >= 4:
        b += [SH_C4[0] * xy * (xx - yy), SH_C4[1] * yz * (3 * xx - yy), SH_C4[2] * xy * (7 * zz - 1),
              SH_C4[3] * yz * (7 * zz - 3), SH_C4[4] * (zz * (35 * zz - 30) + 3),
              SH_C4[5] * xz * (7 * zz - 3), SH_C4[6] * (xx - yy) * (7 * zz - 1),
              SH_C4[7] * xz * (xx - 3 * yy), SH_C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(b, dim=-1)


def rasterize(H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos, sh_degree, means3D, cov3D,
              opacities, shs=None, colors_precomp=None, features=None):
    """Returns (color|None, feature|None, mask(1,H,W), depth(1,H,W), radii). All torch CPU."""
    dt = means3D.dtype
    G = means3D.shape[0]
    vm = viewmatrix.reshape(4, 4).to(dt)  # memory [4*col+row] => vm[col,row]
    pm = projmatrix.reshape(4, 4).to(dt)
    ones = torch.ones(G, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1) @ pm  # (G,4): row-vector convention == transposed matrix
    t = (torch.cat([means3D, ones], 1) @ vm)[:, :3]
    p_w = 1.0 / (ph[:, 3] + 0.0000001)
    ndc = ph[:, :2] * p_w[:, None]
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    inx = (txtz >= -limx) & (txtz <= limx)
    iny = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(inx, t[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(iny, t[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -(fy * ty) / (tz * tz)], -1)], 1)  # (G,2,3)
    Wr = vm[:3, :3].T  # Wr[r][c] = vm_mem[4c+r] = vm[c, r]
    M = J @ Wr
    S = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                     cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], -1).reshape(G, 3, 3)
    cov2 = M @ S @ M.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = (tz > 0.2) & (det != 0)
    det_s = torch.where(det != 0, det, torch.ones_like(det))
    conA, conB, conC = c / det_s, -b / det_s, a / det_s
    with torch.no_grad():
        mid = 0.5 * (a + c)
        disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc)))
    px = (((ndc[:, 0].double() + 1.0) * W - 1.0) * 0.5).to(dt)
    py = (((ndc[:, 1].double() + 1.0) * H - 1.0) * 0.5).to(dt)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        def ti(v):
            return torch.nan_to_num(v / 16.0, nan=0.0, posinf=1e9, neginf=-1e9).to(torch.int64)
        rminx = ti(px - radius).clamp(0, gx)
        rminy = ti(py - radius).clamp(0, gy)
        rmaxx = ti(px + radius + 15).clamp(0, gx)
        rmaxy = ti(py + radius + 15).clamp(0, gy)
        vis = ok & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
        radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    rgb = None
    if shs is not None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=-1, keepdim=True)
        nb = (sh_degree + 1) ** 2
        rgb = torch.einsum("gk,gkc->gc", sh_basis(sh_degree, d), shs[:, :nb]) + 0.5
        rgb = torch.clamp_min(rgb, 0.0)  # autograd: zero grad where clamped (r<0)
    elif colors_precomp is not None:
        rgb = colors_precomp

    # global stable depth order == per-tile order restricted to the tile's members
    with torch.no_grad():
        order = torch.argsort(torch.where(vis, tz, torch.full_like(tz, float("inf"))), stable=True)
        order = order[: int(vis.sum())]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxf, pyf = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)
    tile_x, tile_y = (xs.reshape(-1) // 16), (ys.reshape(-1) // 16)
    o = order
    dx = px[o][None] - pxf[:, None]
    dy = py[o][None] - pyf[:, None]
    power = -0.5 * (conA[o][None] * dx * dx + conC[o][None] * dy * dy) - conB[o][None] * dx * dy
    Gv = torch.exp(torch.clamp(power, max=0.0))
    alpha_raw = opacities.reshape(-1)[o][None] * Gv
    alpha = alpha_raw + (torch.clamp(alpha_raw, max=0.99) - alpha_raw).detach()
    with torch.no_grad():
        in_rect = ((tile_x[:, None] >= rminx[o][None]) & (tile_x[:, None] < rmaxx[o][None]) &
                   (tile_y[:, None] >= rminy[o][None]) & (tile_y[:, None] < rmaxy[o][None]))
        valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        T_excl = torch.cumprod(torch.cat([torch.ones_like(a_eff[:, :1]), 1 - a_eff[:, :-1]], 1), 1)
        stop_here = valid & (T_excl * (1 - a_eff) < 0.0001)
        stopped = torch.cumsum(stop_here.to(torch.int32), 1) > 0
        live = valid & ~stopped
    a_live = torch.where(live, alpha, torch.zeros_like(alpha))
    T = torch.cumprod(torch.cat([torch.ones_like(a_live[:, :1]), 1 - a_live[:, :-1]], 1), 1)
    w = a_live * T
    T_final = T[:, -1] * (1 - a_live[:, -1]) if o.numel() else torch.ones(H * W, dtype=dt)
    color = feat = None
    if rgb is not None:
        color = (w @ rgb[o] + T_final[:, None] * bg.to(dt)[None]).T.reshape(3, H, W)
    if features is not None:
        feat = (w @ features[o]).T.reshape(features.shape[1], H, W)
    mask = (1 - T_final).reshape(1, H, W)
    depth = (w @ tz[o]).reshape(1, H, W)
    return color, feat, mask, depth, radii
