"""Fused image loss (GhOutputs.l1_target, include/gh_raster.h v0.7): mean|image - target| and its gradient from the render
kernel's own epilogue must be what gh_l1_loss computes from the stored image — the gradient bit for bit, the loss up to the
order of its fixed-order float32 sums — and what the L1 term of the reference's loss (utils.py:282-294) is in float64."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _fwd(s, target, **kw):
    from guassianhand_amd.rasterizer import raster_forward
    return raster_forward(s.cams().contiguous(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W,
                          colors_precomp=s.shs.squeeze(1), l1_target=target, **kw)


# (512 x 334 is ragged: the last tile column holds 14 pixels, its last 4x4 blocks 2)
@pytest.mark.parametrize("scene,nv,P", [("random1k", 1, 1000), ("random1k", 3, 2500), ("one_hand", 2, 6000), ("two_hands", 8, None)])
def test_fused_l1_is_gh_l1_loss_of_the_stored_image(dev, scene, nv, P):
    from guassianhand_amd.loss import _l1_kernel
    from guassianhand_amd.scenes import make_scene
    kw = {} if P is None else dict(P=P)
    sc = make_scene(scene, n_views=nv, use_rgb=True, **kw)
    s = sc.to(dev)
    g = torch.Generator().manual_seed(11)
    target = torch.rand(nv, 3, sc.H, sc.W, generator=g).to(dev)
    img, _, ctx = _fwd(s, target)
    assert ctx.l1 is not None
    loss, dimg = ctx.l1
    img0, _, ctx0 = _fwd(s, None)                       # the plain kernel: same image
    assert ctx0.l1 is None and torch.equal(img, img0)
    loss_k, dimg_k = _l1_kernel(img0, target)
    assert torch.equal(dimg, dimg_k)                    # sign(img - gt) / n, bit for bit
    ref = (img0.double() - target.double()).abs().mean().item()
    assert abs(loss.item() - ref) <= 2e-6 * ref and abs(loss_k.item() - ref) <= 2e-6 * ref
    # an exact zero difference has no gradient (torch.abs' backward): make the target equal the render in a patch
    t2 = target.clone()
    t2[:, :, : sc.H // 2, : sc.W // 2] = img0[:, :, : sc.H // 2, : sc.W // 2]
    _, _, ctx2 = _fwd(s, t2)
    assert float(ctx2.l1[1][:, :, : sc.H // 2, : sc.W // 2].abs().max()) == 0.0
    assert torch.equal(ctx2.l1[1], _l1_kernel(img0, t2)[1])
    # bitwise reproducible
    _, _, ctx3 = _fwd(s, target)
    assert torch.equal(ctx3.l1[0], loss) and torch.equal(ctx3.l1[1], dimg)


def test_fused_l1_through_the_one_node_loss_equals_the_two_node_form(dev):
    """loss.rendered_l1_loss takes the fused epilogue; values and every gradient against rasterize_views + l1_mean_loss."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.loss import l1_mean_loss, rendered_l1_loss
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("one_hand", n_views=4, P=8000, use_rgb=True, blend=True)
    s = sc.to(dev)
    cams = s.cams().contiguous()
    names = ("xyz", "opacity", "scaling", "rotation", "shs", "xyz_b", "opacity_b", "color_w", "color_b")
    target = torch.rand(4, 3, sc.H, sc.W, generator=torch.Generator().manual_seed(2)).to(dev)
    pa = {k: getattr(s, k).clone().requires_grad_(True) for k in names}
    pb = {k: getattr(s, k).clone().requires_grad_(True) for k in names}
    kw = lambda p: dict(H=sc.H, W=sc.W, use_rgb=True, xyz_b=p["xyz_b"], opacity_b=p["opacity_b"], color_w=p["color_w"], color_b=p["color_b"])
    img, _ = R.rasterize_views(cams, pa["xyz"], pa["opacity"], pa["scaling"], pa["rotation"], pa["shs"], **kw(pa))
    la = l1_mean_loss(img, target)
    lb, img_b, _ = rendered_l1_loss(cams, pb["xyz"], pb["opacity"], pb["scaling"], pb["rotation"], pb["shs"], target, **kw(pb))
    assert torch.equal(img_b, img.detach())
    assert abs(la.item() - lb.item()) <= 4e-6 * abs(la.item())
    (2.5 * la).backward()
    (2.5 * lb).backward()
    for k in names:
        assert torch.equal(pa[k].grad, pb[k].grad), k


def test_fused_l1_of_an_invalid_call_is_nan_with_no_gradient(dev):
    """An instance overflow poisons the image; the fused loss is then NaN and its gradient zero — gh_l1_loss under its guard."""
    from guassianhand_amd.scenes import make_scene
    s = make_scene("random1k", n_views=2).to(dev)
    target = torch.rand(2, 3, s.H, s.W, generator=torch.Generator().manual_seed(3)).to(dev)
    img, _, ctx = _fwd(s, target, max_instances=128, sync=False)
    torch.cuda.synchronize()
    assert torch.isnan(img).all() and torch.isnan(ctx.l1[0]) and float(ctx.l1[1].abs().max()) == 0.0
    from guassianhand_amd.rasterizer import GhOverflowError, check_overflow
    with pytest.raises(GhOverflowError):
        check_overflow()


def test_fused_l1_of_an_empty_scene_is_the_background(dev):
    from guassianhand_amd.scenes import make_scene
    s = make_scene("random1k", n_views=2, P=1000).to(dev)
    target = torch.rand(2, 3, s.H, s.W, generator=torch.Generator().manual_seed(4)).to(dev)
    from guassianhand_amd.rasterizer import raster_forward
    e = lambda *shape: torch.empty(*shape, device=dev)
    img, _, ctx = raster_forward(s.cams().contiguous(), e(0, 3), e(0, 1), e(0, 3), e(0, 4), H=s.H, W=s.W, colors_precomp=e(0, 3),
                                 l1_target=target)
    ref = (img.double() - target.double()).abs().mean().item()
    assert abs(ctx.l1[0].item() - ref) <= 2e-6 * ref
    inv_n = torch.tensor(1.0 / img.numel(), dtype=torch.float64).to(torch.float32).to(dev)           # (float)(1.0 / n), as the library rounds it
    assert torch.equal(ctx.l1[1], torch.sign(img - target) * inv_n)


def test_fused_l1_combinations_the_library_does_not_fuse(dev):
    """alpha / two streams: ctx.l1 is None on the host side and loss.py runs gh_l1_loss on the image (the C-ABI's own answer,
    GH_ERR_UNSUPPORTED, is tested without a GPU in tests/test_abi.py)."""
    from guassianhand_amd.scenes import make_scene
    s = make_scene("random1k", n_views=2).to(dev)
    target = torch.rand(2, 3, s.H, s.W).to(dev)
    _, _, ctx = _fwd(s, target, return_alpha=True)
    assert ctx.l1 is None
    _, _, ctx = _fwd(s, target, split_streams=True)
    assert ctx.l1 is None


# ---- the fit's image loss (GhOutputs.fit_loss) ----------------------------------------------------------------------------------
def _fit_inputs(sc, nv, dev, seed, with_bbox):
    g = torch.Generator().manual_seed(seed)
    gt_rgb = torch.rand(nv, sc.H, sc.W, 3, generator=g).to(dev)
    gt_mask = (torch.rand(nv, sc.H, sc.W, generator=g) > 0.5).float().to(dev)
    bbox = (torch.rand(nv, sc.H, sc.W, generator=g) > 0.3).float().to(dev) if with_bbox else None
    return gt_rgb, gt_mask, bbox


@pytest.mark.parametrize("scene,nv,P,with_bbox", [("random1k", 2, 2000, False), ("one_hand", 2, 6000, True), ("two_hands", 8, None, True)])
def test_fused_fit_loss_is_gh_fit_loss_of_the_stored_images(dev, scene, nv, P, with_bbox):
    """Full forward and refresh over static lists: gradients w.r.t. image and alpha bit for bit gh_fit_loss's, the loss to the order of
    the float32 sums and against the float64 statement of utils.py:180-252 / :282-294."""
    from guassianhand_amd.loss import _fit_kernel
    from guassianhand_amd.rasterizer import raster_forward
    from guassianhand_amd.scenes import make_scene
    sc = make_scene(scene, n_views=nv, use_rgb=True, **({} if P is None else dict(P=P)))
    s = sc.to(dev)
    gt_rgb, gt_mask, bbox = _fit_inputs(sc, nv, dev, 21, with_bbox)
    spec = (gt_rgb, gt_mask, bbox, 10.0, 1.0, 0.5)
    args = (s.cams().contiguous(), s.xyz, s.opacity, s.scaling, s.rotation)
    kw = dict(H=s.H, W=s.W, colors_precomp=s.shs.squeeze(1), return_alpha=True)
    img, _, ctx = raster_forward(*args, fit_loss=spec, **kw)
    assert ctx.fit is not None and ctx.l1 is None
    loss, dimg, dal = ctx.fit
    img0, _, ctx0 = raster_forward(*args, **kw)
    assert ctx0.fit is None and torch.equal(img, img0) and torch.equal(ctx.alpha, ctx0.alpha)
    loss_k, dimg_k, dal_k = _fit_kernel(img0, ctx0.alpha, *spec)
    assert torch.equal(dimg, dimg_k) and torch.equal(dal, dal_k)
    bb = torch.ones_like(gt_mask) if bbox is None else (bbox != 0).float()
    rgb = (img0 * bb[:, None]).permute(0, 2, 3, 1).double()
    ref = 0.5 * (10.0 * (rgb - gt_rgb.double()).abs().mean(dim=(1, 2, 3)).sum()
                 + ((ctx0.alpha.double().clamp(-0.001, 1.0) - gt_mask.double()) ** 2).mean(dim=(1, 2)).sum()).item()
    assert abs(loss.item() - ref) <= 3e-6 * ref and abs(loss_k.item() - ref) <= 3e-6 * ref
    # static lists + refresh (the fit loop's path): the refresh call fuses it too
    _, _, c_build = raster_forward(*args, static_lists=True, **kw)
    op2 = (s.opacity * 0.8).contiguous()
    img_r, _, c_r = raster_forward(args[0], s.xyz, op2, s.scaling, s.rotation, refresh_of=c_build, fit_loss=spec, **kw)
    assert c_r.fit is not None
    img_p, _, c_p = raster_forward(args[0], s.xyz, op2, s.scaling, s.rotation, **kw)
    assert torch.equal(img_r, img_p) and torch.equal(c_r.alpha, c_p.alpha)
    lk, dk, ak = _fit_kernel(img_p, c_p.alpha, *spec)
    assert torch.equal(c_r.fit[1], dk) and torch.equal(c_r.fit[2], ak) and abs(c_r.fit[0].item() - lk.item()) <= 4e-6 * abs(lk.item())
    # the L1 form over a refresh, too
    tgt = torch.rand(nv, 3, s.H, s.W, generator=torch.Generator().manual_seed(5)).to(dev)
    kw1 = dict(H=s.H, W=s.W, colors_precomp=s.shs.squeeze(1))
    _, _, c_b1 = raster_forward(*args, static_lists=True, **kw1)
    img_r1, _, c_r1 = raster_forward(args[0], s.xyz, op2, s.scaling, s.rotation, refresh_of=c_b1, l1_target=tgt, **kw1)
    assert c_r1.l1 is not None and torch.equal(img_r1, img_p)
    inv_n = torch.tensor(1.0 / img_p.numel(), dtype=torch.float64).to(torch.float32).to(dev)
    assert torch.equal(c_r1.l1[1], torch.sign(img_p - tgt) * inv_n)


def test_fused_fit_loss_through_the_one_node_loss_equals_the_two_node_form(dev):
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.loss import fit_image_loss, rendered_fit_loss
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("one_hand", n_views=3, P=8000, use_rgb=True, blend=True)
    s = sc.to(dev)
    cams = s.cams().contiguous()
    gt_rgb, gt_mask, bbox = _fit_inputs(sc, 3, dev, 31, True)
    names = ("opacity", "shs", "opacity_b", "color_w", "color_b")
    pa = {k: getattr(s, k).clone().requires_grad_(True) for k in names}
    pb = {k: getattr(s, k).clone().requires_grad_(True) for k in names}
    kw = lambda p: dict(H=sc.H, W=sc.W, use_rgb=True, xyz_b=s.xyz_b, opacity_b=p["opacity_b"], color_w=p["color_w"], color_b=p["color_b"])
    img, alpha, _ = R.rasterize_views(cams, s.xyz, pa["opacity"], s.scaling, s.rotation, pa["shs"], return_alpha=True, **kw(pa))
    la = fit_image_loss(img, alpha, gt_rgb, gt_mask, bbox, 10.0, 1.0, 0.125)
    lb, img_b, alpha_b = rendered_fit_loss(cams, s.xyz, pb["opacity"], s.scaling, s.rotation, pb["shs"], gt_rgb, gt_mask, bbox, 10.0, 1.0, 0.125, **kw(pb))
    assert torch.equal(img_b, img.detach()) and torch.equal(alpha_b, alpha.detach())
    assert abs(la.item() - lb.item()) <= 4e-6 * abs(la.item())
    (1.5 * la).backward()
    (1.5 * lb).backward()
    for k in names:
        assert torch.equal(pa[k].grad, pb[k].grad), k


def test_fused_fit_loss_of_an_invalid_call_is_nan_with_no_gradient(dev):
    from guassianhand_amd.rasterizer import GhOverflowError, check_overflow, raster_forward
    from guassianhand_amd.scenes import make_scene
    s = make_scene("random1k", n_views=2).to(dev)
    gt_rgb, gt_mask, bbox = _fit_inputs(s, 2, dev, 41, False)
    img, _, ctx = raster_forward(s.cams().contiguous(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=s.shs.squeeze(1),
                                 return_alpha=True, fit_loss=(gt_rgb, gt_mask, None, 10.0, 1.0, 1.0), max_instances=128, sync=False)
    torch.cuda.synchronize()
    assert torch.isnan(img).all() and torch.isnan(ctx.fit[0]) and float(ctx.fit[1].abs().max()) == 0.0 and float(ctx.fit[2].abs().max()) == 0.0
    with pytest.raises(GhOverflowError):
        check_overflow()


# ---- the final sum inside the backward (GH_FLAG_DEFER_LOSS_SUM / GhGrads.deferred_loss, v0.8) ---------------------------------------
@pytest.mark.parametrize("scene,nv,P", [("random1k", 1, 1000), ("one_hand", 2, 6000), ("two_hands", 8, None)])
def test_deferred_loss_sum_is_the_forward_sum(dev, scene, nv, P):
    """The loss value a spare workgroup of the render backward writes == the one the forward's own sum kernel writes (up to the order
    of two fixed-order float32 sums), the gradients are untouched, and the value is bitwise reproducible. One view runs the
    four-wave backward (the sum by 256 threads), eight views the one-wave form (64 threads)."""
    from guassianhand_amd.rasterizer import raster_backward
    from guassianhand_amd.scenes import make_scene
    kw = {} if P is None else dict(P=P)
    sc = make_scene(scene, n_views=nv, use_rgb=True, **kw)
    s = sc.to(dev)
    target = torch.rand(nv, 3, sc.H, sc.W, generator=torch.Generator().manual_seed(21)).to(dev)
    img0, _, c0 = _fwd(s, target)
    g0 = raster_backward(c0, c0.l1[1])
    assert not c0.defer_loss
    vals = []
    for _ in range(2):
        img1, _, c1 = _fwd(s, target, defer_loss=True)
        assert c1.defer_loss and (c1.dims.flags & 64)
        c1.l1[0].fill_(float("nan"))                     # whatever is there before the backward is not the loss
        g1 = raster_backward(c1, c1.l1[1])
        assert torch.equal(img1, img0) and torch.equal(c1.l1[1], c0.l1[1])
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k
        vals.append(c1.l1[0].clone())
    ref = (img0.double() - target.double()).abs().mean().item()
    assert abs(vals[0].item() - ref) <= 2e-6 * ref and abs(vals[0].item() - c0.l1[0].item()) <= 2e-6 * ref
    assert torch.equal(vals[0], vals[1])


def test_deferred_loss_sum_through_the_one_node_losses(dev):
    """rendered_l1_loss / rendered_fit_loss(defer_loss=True): same gradients as without, the loss tensor holds its value after
    backward(); over static lists (gh_backward_refresh) too; an empty scene (no render backward at all) still gets its sum."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.loss import rendered_fit_loss, rendered_l1_loss
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("one_hand", n_views=3, P=8000, use_rgb=True, blend=True)
    s = sc.to(dev)
    cams = s.cams().contiguous()
    names = ("xyz", "opacity", "scaling", "rotation", "shs", "opacity_b", "color_w", "color_b")
    target = torch.rand(3, 3, sc.H, sc.W, generator=torch.Generator().manual_seed(5)).to(dev)
    gt_rgb, gt_mask, bbox = _fit_inputs(sc, 3, dev, 32, True)
    kw = lambda p: dict(H=sc.H, W=sc.W, use_rgb=True, xyz_b=s.xyz_b, opacity_b=p["opacity_b"], color_w=p["color_w"], color_b=p["color_b"])
    for form in ("l1", "fit", "l1_static", "fit_static"):
        res = []
        for defer in (False, True):
            p = {k: getattr(s, k).clone().requires_grad_(True) for k in names}
            cache = R.GeometryCache() if form.endswith("static") else None
            for _step in range(2 if cache is not None else 1):        # static: build, then a refresh (the second step is the one compared)
                for t in p.values():
                    t.grad = None
                args = (cams, p["xyz"], p["opacity"], p["scaling"], p["rotation"], p["shs"])
                if form.startswith("l1"):
                    loss = rendered_l1_loss(*args, target, geometry_cache=cache, defer_loss=defer, **kw(p))[0]
                else:
                    loss = rendered_fit_loss(*args, gt_rgb, gt_mask, bbox, 10.0, 1.0, 0.5, geometry_cache=cache, defer_loss=defer, **kw(p))[0]
                (1.5 * loss).backward()
            if cache is not None:
                assert cache.hits == 1
            res.append((loss.detach().clone(), {k: p[k].grad.clone() for k in names}))
        (l0, g0), (l1_, g1) = res
        assert abs(l0.item() - l1_.item()) <= 4e-6 * abs(l0.item()), form
        for k in names:
            assert torch.equal(g0[k], g1[k]), (form, k)
    # nothing to draw: gh_backward launches no render kernel, the sum runs on its own
    e = lambda *shape: torch.empty(*shape, device=dev, requires_grad=True)
    tgt = torch.rand(3, 3, sc.H, sc.W, generator=torch.Generator().manual_seed(6)).to(dev)
    loss, img, _ = rendered_l1_loss(cams, e(0, 3), e(0, 1), e(0, 3), e(0, 4), e(0, 1, 3), tgt, H=sc.H, W=sc.W, use_rgb=True, defer_loss=True)
    loss.backward()
    ref = (img.double() - tgt.double()).abs().mean().item()
    assert abs(loss.item() - ref) <= 2e-6 * ref
