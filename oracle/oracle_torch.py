"""Oracle A — dense pure-PyTorch restatement of the rasteriser with autograd-derived gradients.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg). The product
package `guassianhand_amd` never imports this module.

Parity status: "parity unpinned" by the reference — the arithmetic lives in the un-vendored
`diff-gaussian-rasterization` (environment.yml:129; imported at tgs/models/renderer_one_shot.py:3).
This file restates SURVEY.md Appendix A (the published 3DGS algorithm at the API generation used by
tgs/models/renderer_one_shot.py:281-296, :338-346) as a dense pixel x Gaussian expression so that
the *backward* is produced by autograd, independently of the hand-written backward in
oracle/gh_oracle.c and in the HIP kernels. Works in float32 or float64 (finite-difference checks).

Every discrete decision of the forward pass is a constant for autograd (App. A.4-1); the 0.99 alpha
clamp is straight-through (App. A.4-2); the 1.3*tanfov clamp freezes the clamped view-space x/y
(App. A.4-3).
"""
from __future__ import annotations

from typing import Optional

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh (P,M,3), dirs (P,3) unit -> (P,3) (before the +0.5 / clamp)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
               + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
               + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
               + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
               + SH_C3[5] * z * (xx - yy) * sh[:, 14] + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def rasterize_dense(means3D, opacities, scales, rotations, *, viewmatrix, projmatrix, campos,
                    tanfovx: float, tanfovy: float, bg, H: int, W: int,
                    colors_precomp: Optional[torch.Tensor] = None, shs: Optional[torch.Tensor] = None,
                    sh_degree: int = 0, scale_modifier: float = 1.0, pixel_chunk: int = 2048,
                    return_aux: bool = False, pixel_window=None, checkpoint_chunks: bool = False,
                    ambiguity_eps: Optional[float] = None, per_gaussian_only: bool = False,
                    cov3D_precomp: Optional[torch.Tensor] = None, per_gaussian_graph: bool = False):
    """Returns (image (3,H,W), radii (P,) int32[, aux dict]).
    pixel_window = (x0, y0, x1, y1): evaluate only the pixels x0 <= x < x1, y0 <= y < y1 of the H x W image (the image
    returned is (3, y1-y0, x1-x0)); everything else — projection, tile rects, tile membership of a pixel — is that of the
    full image, so the window of a hand-scene render can be checked without the dense P x H x W evaluation.
    checkpoint_chunks: recompute every pixel chunk in the backward instead of keeping its pixel x Gaussian intermediates
    (memory of one chunk at a time).
    ambiguity_eps (with checkpoint_chunks): also return a bool mask (h, w) of the pixels at which some discrete decision of
    App. A.3 sits within that RELATIVE margin of its threshold — alpha against 1/255, power against 0 (|power| <= eps),
    T (1 - alpha) against 1e-4 (margin 5 eps: a product of up to a thousand factors) — i.e. the pixels where an evaluation in
    another precision may legitimately decide differently: returns (image, radii, ambiguous).
    per_gaussian_only: stop after the per-Gaussian stage (App. A.1) — O(P), no pixel is evaluated — and return (None, radii, aux)
    with the projected centre, depth, conic, colour, validity, tile rect and the quantities the discrete decisions are taken on
    (3 sqrt(lambda_max) before the ceil, the rect bounds before the truncation, the SH colour before the clamp).
    per_gaussian_graph (with per_gaussian_only): aux["diff"] additionally holds the stage's continuous outputs WITH their autograd
    graph — px, py, conic (P,3) = (A, B, C), opacity, rgb — so that a caller can push an upstream gradient w.r.t. them back to
    the inputs: the chain rule of App. A.5 as a float64 VJP, O(P), no pixel evaluated (tests/test_gpu_oracle_a_stage.py)."""
    if (shs is None) == (colors_precomp is None):
        raise ValueError("provide exactly one of shs / colors_precomp")
    dt, dev = means3D.dtype, means3D.device
    P = means3D.shape[0]
    V = viewmatrix.to(dt)
    PM = projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt, device=dev)
    mh = torch.cat([means3D, ones], dim=1)
    t = mh @ V                      # row-vector convention (viewmatrix = w2c^T)
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    hom = mh @ PM
    winv = 1.0 / (hom[:, 3] + 1e-7)
    ndcx, ndcy = hom[:, 0] * winv, hom[:, 1] * winv

    if cov3D_precomp is not None:
        # the published module's cov3D_precomp: (P,6) = xx xy xz yy yz zz of Sigma, used as given (no scale_modifier)
        if scales is not None or rotations is not None:
            raise ValueError("provide exactly one of scales / rotations or cov3D_precomp")
        c6 = cov3D_precomp.to(dt)
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]], -1).reshape(P, 3, 3)
    else:
        R = quat_to_rot(rotations)
        Mm = R * (scale_modifier * scales)[:, None, :]
        Sigma = Mm @ Mm.transpose(1, 2)

    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz_safe = torch.where(tz.abs() < 1e-12, torch.full_like(tz, 1e-12), tz)
    txtz, tytz = tx / tz_safe, ty / tz_safe
    xcl = (txtz < -limx) | (txtz > limx)
    ycl = (tytz < -limy) | (tytz > limy)
    cx = torch.where(xcl, (txtz.clamp(-limx, limx) * tz_safe).detach(), tx)
    cy = torch.where(ycl, (tytz.clamp(-limy, limy) * tz_safe).detach(), ty)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -(fx * cx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * cy) / (tz_safe * tz_safe)], dim=-1).reshape(P, 2, 3)
    Wrot = V[:3, :3].transpose(0, 1)          # w2c rotation
    T = J @ Wrot
    cov = T @ Sigma @ T.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    det_safe = torch.where(det == 0, torch.ones_like(det), det)
    cA, cB, cC = c / det_safe, -b / det_safe, a / det_safe
    mid = 0.5 * (a + c)
    sq = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + sq, mid - sq))).detach()
    px = ((ndcx + 1.0) * W - 1.0) * 0.5
    py = ((ndcy + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    with torch.no_grad():
        finite = torch.isfinite(px) & torch.isfinite(py) & torch.isfinite(radius)
        pxs = torch.where(finite, px, torch.zeros_like(px))
        pys = torch.where(finite, py, torch.zeros_like(py))
        rs = torch.where(finite, radius, torch.zeros_like(radius))
        minx = torch.trunc((pxs - rs) / TILE).clamp(0, gx).to(torch.int64)
        miny = torch.trunc((pys - rs) / TILE).clamp(0, gy).to(torch.int64)
        maxx = torch.trunc((pxs + rs + (TILE - 1)) / TILE).clamp(0, gx).to(torch.int64)
        maxy = torch.trunc((pys + rs + (TILE - 1)) / TILE).clamp(0, gy).to(torch.int64)
        tiles = (maxx - minx) * (maxy - miny)
        valid = (tz > 0.2) & (det != 0) & (tiles > 0) & finite
        radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)

    rgb_raw = None
    if colors_precomp is not None:
        rgb = colors_precomp.to(dt)
    else:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb_raw = eval_sh(sh_degree, shs.to(dt), d) + 0.5
        rgb = torch.clamp_min(rgb_raw, 0.0)
    if per_gaussian_only:
        with torch.no_grad():
            r_f = 3.0 * torch.sqrt(torch.maximum(mid + sq, mid - sq))
            edges = torch.stack([(pxs - rs) / TILE, (pys - rs) / TILE, (pxs + rs + (TILE - 1)) / TILE, (pys + rs + (TILE - 1)) / TILE], -1)
            aux = dict(px=px.detach(), py=py.detach(), depth=tz.detach(), conic=torch.stack([cA, cB, cC], -1).detach(), det=det.detach(),
                       cov=torch.stack([a, b, c], -1).detach(), rgb=rgb.detach(), rgb_raw=None if rgb_raw is None else rgb_raw.detach(),
                       valid=valid, rect=torch.stack([minx, miny, maxx, maxy], -1), radius_f=r_f, rect_edges=edges,
                       opacity=opacities.reshape(-1).to(dt).detach())
        if per_gaussian_graph:
            aux["diff"] = dict(px=px, py=py, conic=torch.stack([cA, cB, cC], -1), opacity=opacities.reshape(-1).to(dt), rgb=rgb)
        return None, radii, aux

    # depth order, stable, ties by index (App. A.2); invalid Gaussians pushed to the end
    # The sort key of App. A.2 is the FLOAT32 bit pattern of the depth, whatever precision the rest is evaluated in: a float64
    # run orders by its depth rounded to float32 (ties by index), so that two Gaussians whose depths agree to float32 blend
    # in the order the published algorithm gives them instead of an order only float64 can see.
    with torch.no_grad():
        key = torch.where(valid, tz.to(torch.float32), torch.full_like(tz, float("inf"), dtype=torch.float32))
        order = torch.sort(key, stable=True).indices
    o_px, o_py = px[order], py[order]
    o_A, o_B, o_C = cA[order], cB[order], cC[order]
    o_op = opacities.reshape(-1).to(dt)[order]
    o_rgb = rgb[order]
    o_valid = valid[order]
    o_minx, o_maxx, o_miny, o_maxy = minx[order], maxx[order], miny[order], maxy[order]

    bgc = bg.to(dt)
    wx0, wy0, wx1, wy1 = (0, 0, W, H) if pixel_window is None else [int(v) for v in pixel_window]
    if not (0 <= wx0 < wx1 <= W and 0 <= wy0 < wy1 <= H):
        raise ValueError("pixel_window must lie inside the image")
    ys, xs = torch.meshgrid(torch.arange(wy0, wy1, device=dev), torch.arange(wx0, wx1, device=dev), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    N = xs.numel()
    out_rgb, out_T, out_n = [], [], []

    def chunk_colour(X, Y, o_px, o_py, o_A, o_B, o_C, o_op, o_rgb):
        """Composited colour (n,3) of one pixel chunk — the differentiable part, as a function of the sorted per-Gaussian
        stage so that it can be checkpointed."""
        Xf, Yf = X.to(dt)[:, None], Y.to(dt)[:, None]
        tX, tY = (X // TILE)[:, None], (Y // TILE)[:, None]
        in_tile = (tX >= o_minx[None]) & (tX < o_maxx[None]) & (tY >= o_miny[None]) & (tY < o_maxy[None]) & o_valid[None]
        dx = o_px[None, :] - Xf
        dy = o_py[None, :] - Yf
        power = -0.5 * (o_A[None] * dx * dx + o_C[None] * dy * dy) - o_B[None] * dx * dy
        power = torch.where(in_tile, power, torch.zeros_like(power))
        a_raw = o_op[None, :] * torch.exp(torch.clamp(power, max=0.0))
        alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
        with torch.no_grad():
            contrib0 = in_tile & (power <= 0) & (alpha >= 1.0 / 255.0)
        one_minus = torch.where(contrib0, 1.0 - alpha, torch.ones_like(alpha))
        T_incl = torch.cumprod(one_minus, dim=1)
        with torch.no_grad():
            active = contrib0 & (T_incl >= 1e-4)
        T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
        wgt = torch.where(active, alpha * T_excl, torch.zeros_like(alpha))
        final_T = torch.prod(torch.where(active, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
        col = wgt @ o_rgb + final_T[:, None] * bgc[None, :]
        if ambiguity_eps is None:
            return col
        with torch.no_grad():
            e = float(ambiguity_eps)
            thr = 1.0 / 255.0
            near_alpha = in_tile & (power <= e) & ((a_raw - thr).abs() <= e * thr)
            near_power = in_tile & (power.abs() <= e) & (o_op[None, :] >= thr * (1 - e))
            alive = torch.cat([torch.ones_like(T_incl[:, :1], dtype=torch.bool), T_incl[:, :-1] >= 1e-4 * (1 - 5 * e)], dim=1)
            near_stop = contrib0 & alive & ((T_incl - 1e-4).abs() <= 5 * e * 1e-4)
            # a decision behind the stop cannot matter: only entries the walk can still reach count
            amb = ((near_alpha | near_power) & alive | near_stop).any(dim=1)
        return col, amb.to(col.dtype)

    if checkpoint_chunks and not return_aux:
        from torch.utils.checkpoint import checkpoint
        amb_parts = []
        for s in range(0, N, pixel_chunk):
            r = checkpoint(chunk_colour, xs[s:s + pixel_chunk], ys[s:s + pixel_chunk], o_px, o_py, o_A, o_B, o_C, o_op,
                           o_rgb, use_reentrant=False)
            if ambiguity_eps is not None:
                r, am = r
                amb_parts.append(am.detach() > 0)
            out_rgb.append(r)
        img = torch.cat(out_rgb, dim=0).reshape(wy1 - wy0, wx1 - wx0, 3).permute(2, 0, 1).contiguous()
        if ambiguity_eps is not None:
            return img, radii, torch.cat(amb_parts).reshape(wy1 - wy0, wx1 - wx0)
        return img, radii

    for s in range(0, N, pixel_chunk):
        X = xs[s:s + pixel_chunk]
        Y = ys[s:s + pixel_chunk]
        Xf, Yf = X.to(dt)[:, None], Y.to(dt)[:, None]
        tX, tY = (X // TILE)[:, None], (Y // TILE)[:, None]
        in_tile = (tX >= o_minx[None]) & (tX < o_maxx[None]) & (tY >= o_miny[None]) & (tY < o_maxy[None]) & o_valid[None]
        dx = o_px[None, :] - Xf
        dy = o_py[None, :] - Yf
        power = -0.5 * (o_A[None] * dx * dx + o_C[None] * dy * dy) - o_B[None] * dx * dy
        power = torch.where(in_tile, power, torch.zeros_like(power))   # keep masked lanes finite
        a_raw = o_op[None, :] * torch.exp(torch.clamp(power, max=0.0))
        alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()      # straight-through clamp
        with torch.no_grad():
            contrib0 = in_tile & (power <= 0) & (alpha >= 1.0 / 255.0)
        one_minus = torch.where(contrib0, 1.0 - alpha, torch.ones_like(alpha))
        T_incl = torch.cumprod(one_minus, dim=1)
        with torch.no_grad():
            active = contrib0 & (T_incl >= 1e-4)
        T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
        wgt = torch.where(active, alpha * T_excl, torch.zeros_like(alpha))
        C = wgt @ o_rgb
        final_T = torch.prod(torch.where(active, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
        out_rgb.append(C + final_T[:, None] * bgc[None, :])
        if return_aux:
            with torch.no_grad():
                # n_contrib: 1-based position of the last blended entry *within the pixel's tile list*
                pos_in_tile = torch.cumsum(in_tile.to(torch.int64), dim=1)
                last = torch.where(active, pos_in_tile, torch.zeros_like(pos_in_tile)).max(dim=1).values
                out_T.append(final_T.detach())
                out_n.append(last)
    img = torch.cat(out_rgb, dim=0).reshape(wy1 - wy0, wx1 - wx0, 3).permute(2, 0, 1).contiguous()
    if not return_aux:
        return img, radii
    aux = dict(px=px.detach(), py=py.detach(), depth=tz.detach(), conic=torch.stack([cA, cB, cC], -1).detach(),
               rgb=rgb.detach(), valid=valid, rect=torch.stack([minx, miny, maxx, maxy], -1),
               final_T=torch.cat(out_T).reshape(wy1 - wy0, wx1 - wx0), n_contrib=torch.cat(out_n).reshape(wy1 - wy0, wx1 - wx0),
               num_rendered=int(tiles[valid].sum().item()))
    return img, radii, aux


def blend_attributes(xyz, opacity, shs, *, use_rgb: bool, color_w=None, xyz_b=None, color_b=None, opacity_b=None):
    """Restatement of the attribute blend at tgs/models/renderer_one_shot.py:298-334 (torch ops in the
    reference's own order, incl. the SH-mode double multiply when color_b is given)."""
    means3D = xyz if xyz_b is None else xyz + xyz_b
    op = opacity if opacity_b is None else opacity + opacity_b.view(-1, 1)
    colors_precomp, out_shs = None, None
    if use_rgb:
        colors_precomp = shs.squeeze(1)
        if color_w is not None:
            w = color_w.view(-1, 16, 3)
            colors_precomp = colors_precomp * w[:, 0, :] + w[:, 1, :] - 1
        if color_b is not None:
            colors_precomp = colors_precomp + color_b.view(-1, 16, 3)[:, 0, :]
    else:
        out_shs = shs
        if color_w is not None:
            out_shs = out_shs * color_w.view(-1, 16, 3)
        if color_b is not None:
            out_shs = out_shs * color_w.view(-1, 16, 3) + color_b.view(-1, 16, 3)
    return means3D, op, colors_precomp, out_shs


def select_gaussians_reference(if_gs_valid, query_points, gs_hidden_features, threshold_low: float = 0.1, threshold_high: float = 0.9):
    """Restatement of tgs/models/renderer_one_shot.py:468-473 (forward_single_batch): the validity prune and the rows that
    are duplicated, by boolean-mask indexing exactly as the reference writes it. Returns (query_points_valid,
    gs_hidden_features_valid, query_points_copied, gs_hidden_features_copied)."""
    s = if_gs_valid.squeeze(1)
    query_points_valid = query_points[s > threshold_low]
    gs_hidden_features_valid = gs_hidden_features[s > threshold_low]
    query_points_copied = query_points[s > threshold_high]
    gs_hidden_features_copied = gs_hidden_features[s > threshold_high]
    return query_points_valid, gs_hidden_features_valid, query_points_copied, gs_hidden_features_copied
