"""The drop-in surface: `GaussianRasterizationSettings` / `GaussianRasterizer` called exactly as
tgs/models/renderer_one_shot.py:281-296, :338-346, :372-379 call them, autograd included, and the
`forward_single_view` / view-batched mirror of the reference's render loop."""
import math

import pytest
import torch

from tests.helpers import forward_single_view, max_rel, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _reference_style_call(sc, dev, use_shim=False):
    if use_shim:
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    else:
        from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from guassianhand_amd.camera import Camera
    s = sc.to(dev)
    cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W, 0.71, 1.42)   # built on the host like the oracle's record
    for a in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, a, getattr(cam, a).to(dev))
    xyz = s.xyz.clone().requires_grad_(True)
    opacity, scales, rots = s.opacity.clone().requires_grad_(True), s.scaling.clone().requires_grad_(True), s.rotation.clone().requires_grad_(True)
    col = s.shs.clone().requires_grad_(True)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=dev) + 0
    screenspace_points.retain_grad()
    rs = GaussianRasterizationSettings(
        image_height=int(cam.height), image_width=int(cam.width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=s.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform.float(), sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=rs)
    with torch.autocast(device_type="cuda", dtype=torch.float32):
        if sc.use_rgb:
            img, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=col.squeeze(1),
                                    opacities=opacity, scales=scales, rotations=rots, cov3D_precomp=None)
        else:
            img, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=col, colors_precomp=None,
                                    opacities=opacity, scales=scales, rotations=rots, cov3D_precomp=None)
    from guassianhand_amd.camera import pack_camera
    cam_rec = pack_camera(rs.viewmatrix, rs.projmatrix, rs.campos, rs.tanfovx, rs.tanfovy, rs.bg).cpu()
    return img, radii, dict(cam_rec=cam_rec, means3D=xyz, opacities=opacity, scales=scales, rotations=rots, col=col, means2D=screenspace_points)


@pytest.mark.parametrize("use_rgb", [True, False])
def test_reference_call_protocol_with_autograd(dev, use_rgb):
    from guassianhand_amd.scenes import make_scene
    from oracle.oracle_c import OracleRender
    sc = make_scene("random1k", n_views=1, use_rgb=use_rgb)
    img, radii, leaves = _reference_style_call(sc, dev, use_shim=True)
    assert img.shape == (3, sc.H, sc.W) and img.dtype == torch.float32 and img.is_contiguous()
    assert radii.shape == (sc.P,) and radii.dtype == torch.int32
    gt = torch.rand(3, sc.H, sc.W, generator=torch.Generator().manual_seed(1)).to(dev)
    loss = (img - gt).abs().mean()
    loss.backward()
    kw = dict(colors_precomp=sc.shs.squeeze(1)) if use_rgb else dict(shs=sc.shs, sh_degree=3)
    o = OracleRender(leaves["cam_rec"], sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, **kw)
    assert torch.equal(img.detach().cpu(), o.image[0])
    dimg = (torch.sign(o.image[0] - gt.cpu()) / gt.numel())[None]
    og = o.backward(dimg)
    for k, ok in (("means3D", "means3D"), ("opacities", "opacities"), ("scales", "scales"), ("rotations", "rotations"),
                  ("col", "colors_precomp" if use_rgb else "shs"), ("means2D", "means2D")):
        a, b = leaves[k].grad.cpu(), og[ok]
        assert a.reshape(-1).shape == b.reshape(-1).shape
        assert max_rel(a, b) <= 1e-3 and rel_l2(a, b) <= 1e-5, k
    # means2D.grad layout of App. A.4-8: z component is zero
    assert float(leaves["means2D"].grad[:, 2].abs().max()) == 0.0


def test_forward_single_view_equals_fused_batched_views(dev):
    """Reference protocol (torch blend + 2 rasteriser calls per view, renderer_one_shot.py:259-382) vs the
    MI355X form (all views in one launch sequence, blend fused into the kernels): same images, same grads."""
    from guassianhand_amd import renderer as R
    from guassianhand_amd.camera import Camera, pack_cameras_from_w2c
    from guassianhand_amd.scenes import make_scene
    for use_rgb in (True, False):
        sc = make_scene("random1k", n_views=3, P=1500, use_rgb=use_rgb, blend=True)
        s = sc.to(dev)
        names = ("xyz", "opacity", "rotation", "scaling", "shs", "color_w", "color_b", "opacity_b", "xyz_b")
        gt = torch.rand(3, sc.H, sc.W, 3, generator=torch.Generator().manual_seed(2)).to(dev)

        def leaves():
            return {n: getattr(s, n).clone().requires_grad_(True) for n in names}

        a = leaves()
        gs = R.GaussianModel(a["xyz"], a["opacity"], a["rotation"], a["scaling"], a["shs"])
        outs = []
        for v in range(3):
            cam = Camera.from_w2c(s.w2c[v], s.K[v], sc.H, sc.W, 0.71, 1.42)
            rec = pack_cameras_from_w2c(s.w2c[v:v + 1], s.K[v:v + 1], sc.H, sc.W, s.bg)[0]   # same device maths as render_views
            cam.world_view_transform, cam.full_proj_transform = rec[:16].reshape(4, 4), rec[16:32].reshape(4, 4)
            cam.camera_center = rec[32:35]
            cam.FoVx, cam.FoVy = 2 * torch.atan(rec[35]), 2 * torch.atan(rec[36])
            outs.append(forward_single_view(gs, cam, s.bg, color_w=a["color_w"], xyz_b=a["xyz_b"], color_b=a["color_b"],
                                              opacity_b=a["opacity_b"], use_rgb=use_rgb, sh_degree=3))
        rgb_a = torch.stack([o["comp_rgb"] for o in outs])
        mask_a = torch.stack([o["comp_mask"] for o in outs])
        ((rgb_a - gt).abs().mean() + ((mask_a.mean(-1) - gt[..., 0]) ** 2).mean()).backward()

        b = leaves()
        gsb = R.GaussianModel(b["xyz"], b["opacity"], b["rotation"], b["scaling"], b["shs"])
        ob = R.render_views(gsb, s.w2c, s.K, sc.H, sc.W, s.bg, color_w=b["color_w"], xyz_b=b["xyz_b"], color_b=b["color_b"],
                            opacity_b=b["opacity_b"], use_rgb=use_rgb, sh_degree=3)
        assert ob["comp_rgb"].shape == (3, sc.H, sc.W, 3)
        ((ob["comp_rgb"] - gt).abs().mean() + ((ob["comp_mask"].mean(-1) - gt[..., 0]) ** 2).mean()).backward()
        assert torch.equal(rgb_a, ob["comp_rgb"]) and torch.equal(mask_a, ob["comp_mask"])
        for n in names:
            assert max_rel(b[n].grad.cpu(), a[n].grad.cpu()) <= 1e-3 and rel_l2(b[n].grad.cpu(), a[n].grad.cpu()) <= 2e-5, n


def test_fused_alpha_channel_equals_separate_mask_pass(dev):
    """the mask render of renderer_one_shot.py:353-380 (colour 1, bg 0, second rasteriser call)
    comes out of the SAME walk as a 4th channel: bit-identical forward, matching gradients through both outputs."""
    from guassianhand_amd.rasterizer import raster_backward, raster_forward
    from guassianhand_amd.scenes import make_scene
    from oracle.oracle_c import OracleRender
    for cfg, kw in (("random1k", dict(P=3000, n_views=2)), ("one_hand", dict(P=20000, n_views=1))):
        sc = make_scene(cfg, blend=True, **kw)
        s = sc.to(dev)
        cams = sc.cams().to(dev)
        bl = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
        args = (cams, s.xyz, s.opacity, s.scaling, s.rotation)
        img, _, ctx = raster_forward(*args, H=sc.H, W=sc.W, colors_precomp=s.shs.squeeze(1), return_alpha=True, **bl)
        mcams = cams.clone(); mcams[:, 37:40] = 0
        mask, _, mctx = raster_forward(mcams, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W,
                                       colors_precomp=torch.ones_like(s.xyz), xyz_b=s.xyz_b, opacity_b=s.opacity_b)
        assert torch.equal(ctx.alpha, mask[:, 0]) and torch.equal(mask[:, 0], mask[:, 2])
        # the oracle's separate mask render agrees bit for bit as well
        o = OracleRender(mcams.cpu(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W,
                         colors_precomp=torch.ones_like(sc.xyz), xyz_b=sc.xyz_b, opacity_b=sc.opacity_b)
        assert torch.equal(ctx.alpha.cpu(), o.image[:, 0])
        g = torch.Generator().manual_seed(3)
        d_img = torch.randn(img.shape, generator=g).to(dev)
        d_a = torch.randn(ctx.alpha.shape, generator=g).to(dev)
        fused = raster_backward(ctx, d_img, dL_dalpha=d_a)
        g_rgb = raster_backward(ctx, d_img)
        g_mask = raster_backward(mctx, d_a[:, None].expand(-1, 3, -1, -1).contiguous() / 1.0 * torch.tensor([1.0, 0.0, 0.0], device=dev)[None, :, None, None])
        for k in ("means3D", "opacities", "scales", "rotations", "xyz_b", "opacity_b"):
            want = g_rgb[k] + g_mask[k]
            assert max_rel(fused[k].cpu(), want.cpu()) <= 1e-3 and rel_l2(fused[k].cpu(), want.cpu()) <= 1e-5, k
        for k in ("colors_precomp", "color_w", "color_b"):      # the mask has no colour dependence
            assert torch.equal(fused[k], g_rgb[k]), k
        og = o.backward(d_a[:, None].cpu() * torch.tensor([1.0, 0.0, 0.0])[None, :, None, None])
        only_alpha = raster_backward(ctx, None, dL_dalpha=d_a)
        for k in ("means3D", "opacities", "scales", "rotations"):
            assert max_rel(only_alpha[k].cpu(), og[k]) <= 1e-3 and rel_l2(only_alpha[k].cpu(), og[k]) <= 1e-5, k


def test_mask_pass_is_accumulated_alpha(dev):
    """The mask render (colour = 1, bg = 0, renderer_one_shot.py:353-380) equals 1 - final_T."""
    from guassianhand_amd.rasterizer import raster_forward, workspace_views
    from guassianhand_amd.scenes import make_scene
    s = make_scene("one_hand", n_views=1, P=8000).to(dev)
    img, _, ctx = raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W,
                                 colors_precomp=torch.ones_like(s.xyz))
    T = workspace_views(ctx)["final_T"]
    assert (img[0, 0] - (1 - T[0])).abs().max() < 2e-6
    assert torch.equal(img[0, 0], img[0, 1]) and torch.equal(img[0, 1], img[0, 2])


def test_rgb_compact_color_b_equals_padded_reference_layout(dev):
    """GH_FLAG_BLEND_COLOR_B_RGB: with colors_precomp the blend reads color_b.view(-1,16,3)[:,0,:] only
    (renderer_one_shot.py:328), so a (P,3) tensor of those columns must give the image and gradients of the (P,48) one."""
    from guassianhand_amd.rasterizer import rasterize_views
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=2500, blend=True).to(dev)
    cams = sc.cams()
    outs = []
    for compact in (False, True):
        cb = (sc.color_b[:, :3].contiguous() if compact else sc.color_b.clone()).requires_grad_(True)
        xyz = sc.xyz.clone().requires_grad_(True)
        img, _ = rasterize_views(cams, xyz, sc.opacity, sc.scaling, sc.rotation, sc.shs, H=sc.H, W=sc.W, use_rgb=True,
                                 xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=cb)
        (img * torch.linspace(0, 1, img.numel(), device=dev).view_as(img)).sum().backward()
        outs.append((img.detach(), xyz.grad, cb.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[1][2].shape == (sc.P, 3) and torch.equal(outs[0][2][:, :3], outs[1][2])
    assert float(outs[0][2][:, 3:].abs().max()) == 0.0
    with pytest.raises(ValueError):
        rasterize_views(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, sc.shs, H=sc.H, W=sc.W, use_rgb=True,
                        color_w=sc.color_w, color_b=sc.color_b[:, :5].contiguous())


@pytest.mark.parametrize("use_rgb", [True, False])
def test_pose_batch_equals_per_item_renders(dev, use_rgb):
    """GH_FLAG_PER_VIEW_GAUSSIANS: a batch of DIFFERENT Gaussian sets (the batch loop of GS3DRenderer.forward,
    renderer_one_shot.py:615-633), one camera each, in one launch sequence == rendering every item on its own:
    images bit-identical, per-Gaussian gradients identical, shared blend parameters (color_w, xyz_b) summed."""
    from guassianhand_amd.rasterizer import rasterize_views
    from guassianhand_amd.scenes import make_scene
    B, P = 3, 1500
    items = [make_scene("random1k", n_views=1, P=P, use_rgb=use_rgb, blend=True, seed=100 + b).to(dev) for b in range(B)]
    ref = items[0]
    H, W = ref.H, ref.W
    cams = torch.cat([it.cams() for it in items])
    color_w, xyz_b = ref.color_w, torch.tensor([0.004, -0.003, 0.002], device=dev)
    dimg = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    names = ("xyz", "opacity", "scaling", "rotation", "shs", "opacity_b", "color_b")

    def leaves(ts):
        return [t.clone().requires_grad_(True) for t in ts]

    # per item
    per_imgs, per_grads, gw, gx = [], [], 0, 0
    for b, it in enumerate(items):
        xs = leaves([getattr(it, k) for k in names])
        cw, xb = leaves([color_w, xyz_b])
        img, _ = rasterize_views(cams[b:b + 1], xs[0], xs[1], xs[2], xs[3], xs[4], H=H, W=W, use_rgb=use_rgb, sh_degree=it.sh_degree,
                                 xyz_b=xb, opacity_b=xs[5], color_w=cw, color_b=xs[6])
        (img * dimg[b:b + 1]).sum().backward()
        per_imgs.append(img.detach()); per_grads.append([x.grad for x in xs]); gw = gw + cw.grad; gx = gx + xb.grad
    # as one pose batch
    xs = leaves([torch.cat([getattr(it, k) for it in items]) for k in names])
    cw, xb = leaves([color_w, xyz_b])
    img, radii = rasterize_views(cams, xs[0], xs[1], xs[2], xs[3], xs[4], H=H, W=W, use_rgb=use_rgb, sh_degree=ref.sh_degree,
                                 xyz_b=xb, opacity_b=xs[5], color_w=cw, color_b=xs[6], per_view_gaussians=True)
    (img * dimg).sum().backward()
    assert radii.shape == (B, P)
    assert torch.equal(img.detach(), torch.cat(per_imgs))
    for j, k in enumerate(names):
        want = torch.cat([g[j] for g in per_grads])
        assert torch.equal(xs[j].grad, want), k
    assert torch.allclose(cw.grad, gw, rtol=1e-4, atol=1e-6) and torch.allclose(xb.grad, gx, rtol=1e-4, atol=1e-6)


def test_rendered_loss_is_the_two_node_form_bit_for_bit(dev):
    """Render + loss as one autograd node (GhGrads.upstream_scale applies dL/dloss inside the kernel) against
    rasterize_views followed by the loss Function, with a non-trivial upstream factor."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.loss import fit_image_loss, l1_mean_loss, rendered_fit_loss, rendered_l1_loss
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=3, P=2500, use_rgb=True, blend=True)
    s = sc.to(dev)
    cams = s.cams().contiguous()
    names = ("xyz", "opacity", "scaling", "rotation", "shs", "xyz_b", "opacity_b", "color_w", "color_b")
    g = torch.Generator().manual_seed(5)
    target = torch.rand(3, 3, sc.H, sc.W, generator=g).to(dev)
    gt_rgb, gt_mask = torch.rand(3, sc.H, sc.W, 3, generator=g).to(dev), (torch.rand(3, sc.H, sc.W, generator=g) > 0.5).float().to(dev)

    def leaves():
        return {k: getattr(s, k).clone().requires_grad_(True) for k in names}

    def args(p):
        return (cams, p["xyz"], p["opacity"], p["scaling"], p["rotation"], p["shs"]), dict(
            H=sc.H, W=sc.W, use_rgb=True, xyz_b=p["xyz_b"], opacity_b=p["opacity_b"], color_w=p["color_w"], color_b=p["color_b"])

    for kind in ("l1", "fit"):
        pa, pb = leaves(), leaves()
        a_pos, a_kw = args(pa)
        b_pos, b_kw = args(pb)
        if kind == "l1":
            img, _ = R.rasterize_views(*a_pos, **a_kw)
            la = l1_mean_loss(img, target)
            lb, img_b, _ = rendered_l1_loss(*b_pos, target, **b_kw)
        else:
            img, alpha, _ = R.rasterize_views(*a_pos, return_alpha=True, **a_kw)
            la = fit_image_loss(img, alpha, gt_rgb, gt_mask, None, 10.0, 1.0, 0.5)
            lb, img_b, alpha_b = rendered_fit_loss(*b_pos, gt_rgb, gt_mask, None, 10.0, 1.0, 0.5, **b_kw)
            assert torch.equal(alpha_b, alpha.detach())
        # (the one-node forms take the loss from the render kernel's epilogue, GhOutputs.l1_* / fit_loss: the same gradients bit for bit,
        # the loss itself up to the order of its float32 sums — tests/test_gpu_fused_l1.py)
        assert abs(la.item() - lb.item()) <= 4e-6 * abs(la.item())
        assert torch.equal(img_b, img.detach())
        (3.0 * la).backward()
        (3.0 * lb).backward()
        for k in names:
            assert torch.equal(pa[k].grad, pb[k].grad), (kind, k)


@pytest.mark.parametrize("use_rgb", [True, False])
def test_second_call_over_the_same_geometry_reuses_the_first_calls_lists_bit_for_bit(dev, use_rgb):
    """the reference's mask pass (renderer_one_shot.py:372-379) follows the RGB pass (:338-346) with the
    same means3D / opacity / scales / rotations / camera objects. The drop-in recognises that and only re-walks the first
    call's tile lists with the new colours (gh_forward_shared / gh_backward_shared). Images, radii and every gradient must
    equal the plain two-full-calls protocol bit for bit; a call with different geometry must NOT be matched."""
    from guassianhand_amd import rasterizer as Rz
    from guassianhand_amd import renderer as R
    from guassianhand_amd.camera import Camera
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=3000, use_rgb=use_rgb, blend=True)
    s = sc.to(dev)
    names = ("xyz", "opacity", "rotation", "scaling", "shs", "color_w", "color_b", "opacity_b", "xyz_b")
    gt = torch.rand(2, sc.H, sc.W, 3, generator=torch.Generator().manual_seed(7)).to(dev)
    cams = [Camera.from_w2c(s.w2c[v], s.K[v], sc.H, sc.W) for v in range(2)]

    def run(reuse):
        Rz.set_geometry_reuse(reuse)
        a = {n: getattr(s, n).clone().requires_grad_(True) for n in names}
        gs = R.GaussianModel(a["xyz"], a["opacity"], a["rotation"], a["scaling"], a["shs"])
        outs = [forward_single_view(gs, cams[v], s.bg, color_w=a["color_w"], xyz_b=a["xyz_b"], color_b=a["color_b"],
                                      opacity_b=a["opacity_b"], use_rgb=use_rgb, sh_degree=3) for v in range(2)]
        rgb = torch.stack([o["comp_rgb"] for o in outs]); mask = torch.stack([o["comp_mask"] for o in outs])
        ((rgb - gt).abs().mean() + ((mask.mean(-1) - gt[..., 0]) ** 2).mean()).backward()
        return rgb.detach(), mask.detach(), {n: a[n].grad for n in names}

    try:
        rgb0, mask0, g0 = run(False)
        rgb1, mask1, g1 = run(True)
    finally:
        Rz.set_geometry_reuse(True)
    assert torch.equal(rgb0, rgb1) and torch.equal(mask0, mask1)
    assert float(mask1.max()) > 0.5                                      # the mask pass really rendered something
    for n in names:
        assert torch.equal(g0[n], g1[n]), n
    # the second call of a view really took the shared path (its context has a parent), a different view did not
    from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    import math
    cam = cams[0]
    mk = lambda bg: GaussianRasterizationSettings(
        image_height=sc.H, image_width=sc.W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
        scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform.float(), sh_degree=0,
        campos=cam.camera_center, prefiltered=False, debug=False)
    xyz = s.xyz.clone().requires_grad_(True)
    col = torch.rand(sc.P, 3, device=dev)
    kw = dict(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=s.opacity, scales=s.scaling, rotations=s.rotation, cov3D_precomp=None)
    img_a, _ = GaussianRasterizer(mk(s.bg))(colors_precomp=col, **kw)
    img_b, _ = GaussianRasterizer(mk(torch.zeros(3, device=dev)))(colors_precomp=torch.ones_like(col), **kw)
    assert img_a.grad_fn.rctx.parent is None and img_b.grad_fn.rctx.parent is img_a.grad_fn.rctx
    assert Rz._geom_last is None                                          # the record is consumed by its one reuse ...
    img_b2, _ = GaussianRasterizer(mk(s.bg))(colors_precomp=col, **kw)
    assert img_b2.grad_fn.rctx.parent is None                             # ... so a third call over the same objects is a full call
    kw2 = dict(kw, opacities=s.opacity.clone())                           # another tensor object: geometry not provably the same
    img_c, _ = GaussianRasterizer(mk(s.bg))(colors_precomp=col, **kw2)
    assert img_c.grad_fn.rctx.parent is None
    objs = Rz._geom_last[0]                                               # weak references only: the record keeps no tensor alive
    del kw2, img_c                                                        # (the call's own context held the tensor until now)
    import gc
    gc.collect()
    assert any(o() is None for o in objs)


def _dropin_leaves(sc, dev):
    import math
    from guassianhand_amd.camera import Camera
    from guassianhand_amd.rasterizer import GaussianRasterizationSettings
    s = sc.to(dev)
    cam = Camera.from_w2c(s.w2c[0], s.K[0], sc.H, sc.W)
    rs = GaussianRasterizationSettings(
        image_height=sc.H, image_width=sc.W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=s.bg,
        scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform.float(), sh_degree=0,
        campos=cam.camera_center, prefiltered=False, debug=False)
    xyz = s.xyz.clone().requires_grad_(True)
    col = s.shs.squeeze(1).clone().requires_grad_(True)
    kw = dict(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=s.opacity, scales=s.scaling, rotations=s.rotation,
              colors_precomp=col, cov3D_precomp=None)
    return rs, kw, xyz, col


def test_default_dropin_never_lets_an_overflowed_render_reach_the_optimiser(dev):
    """with the default sync=None a call under autograd is sync-free after the first of its shape. If its
    instance count then exceeds the learned capacity, the image is NaN (device-side guard) — and the call's BACKWARD raises
    GhOverflowError before any gradient exists, so a reference-style loop never steps its optimiser on NaN. The re-run fits."""
    from guassianhand_amd import rasterizer as Rz
    from guassianhand_amd.rasterizer import GaussianRasterizer
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=3100)
    rs, kw, xyz, col = _dropin_leaves(sc, dev)
    Rz.check_overflow()
    img0, _ = GaussianRasterizer(rs)(**kw)                                # first call of the shape: reads D, learns the capacity
    img0.sum().backward()
    g_ref = xyz.grad.clone()
    xyz.grad = None; col.grad = None
    key = Rz.capacity_key(sc.P, 1, sc.H, sc.W, False)
    D = Rz.last_num_rendered()
    assert D > 2048 and key in Rz._capacity
    Rz._capacity[key] = D // 2                                            # as if the scene had grown since the capacity was learned
    img1, _ = GaussianRasterizer(rs)(**kw)                                # sync-free: no error here
    assert torch.isnan(img1).all()
    with pytest.raises(Rz.GhOverflowError):
        img1.sum().backward()
    assert xyz.grad is None and col.grad is None                          # nothing reached the leaves
    assert Rz._capacity[key] >= D                                         # capacity raised: the re-run step is whole
    img2, _ = GaussianRasterizer(rs)(**kw)
    img2.sum().backward()
    assert torch.equal(img2, img0) and torch.equal(xyz.grad, g_ref)
    Rz.check_overflow()


def test_default_dropin_inference_call_reruns_transparently(dev):
    """... and a call no backward will follow (no input requires grad / torch.no_grad: an inference render) reads D back
    like the reference wrapper and is re-run with a larger capacity: it can never return a NaN image silently."""
    from guassianhand_amd import rasterizer as Rz
    from guassianhand_amd.rasterizer import GaussianRasterizer
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=3200)
    rs, kw, xyz, col = _dropin_leaves(sc, dev)
    Rz.set_geometry_reuse(False)                   # (every call below is to be a full call, not the mask pass of the one before)
    try:
        img0, _ = GaussianRasterizer(rs)(**kw)
        key = Rz.capacity_key(sc.P, 1, sc.H, sc.W, False)
        D = Rz.last_num_rendered()
        Rz._capacity[key] = D // 2
        with torch.no_grad():
            img1, _ = GaussianRasterizer(rs)(**kw)
        assert torch.equal(img1, img0.detach()) and Rz._capacity[key] >= D
        Rz._capacity[key] = D // 2
        kw_ng = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in kw.items()}
        img2, _ = GaussianRasterizer(rs)(**kw_ng)
        assert torch.equal(img2, img0.detach())
        Rz.check_overflow()
    finally:
        Rz.set_geometry_reuse(True)


def test_backward_after_an_inplace_update_of_an_input_raises_like_the_reference(dev):
    """The backward kernels read the call's input tensors again (through pointers the context keeps). The reference extension saves
    its inputs with `save_for_backward`, so PyTorch raises when one of them was written in place between forward and backward; the
    drop-in, the view-batched Function and the render+loss node register theirs too — the same error instead of gradients that mix
    the old tile lists with the new values. A write to a tensor the call did not read is no error."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.loss import rendered_l1_loss
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=600, use_rgb=True, blend=False)
    img, _, t = _reference_style_call(sc, dev)
    with torch.no_grad():
        t["scales"].mul_(1.01)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        img.sum().backward()
    # an untouched call still works afterwards, and its gradients are the oracle-checked ones of the other tests
    img, _, t = _reference_style_call(sc, dev)
    with torch.no_grad():
        t["means2D"].add_(1.0)                               # never read by the rasteriser
    img.sum().backward()
    assert t["means3D"].grad is not None and bool(torch.isfinite(t["means3D"].grad).all())

    s = sc.to(dev)
    cams = s.cams().contiguous()
    target = torch.rand(2, 3, sc.H, sc.W, device=dev)
    for form in ("views", "loss"):
        p = {k: getattr(s, k).clone().requires_grad_(True) for k in ("xyz", "opacity", "scaling", "rotation", "shs")}
        pos = (cams, p["xyz"], p["opacity"], p["scaling"], p["rotation"], p["shs"])
        if form == "views":
            out = R.rasterize_views(*pos, H=sc.H, W=sc.W, use_rgb=True)[0].sum()
        else:
            out = rendered_l1_loss(*pos, target, H=sc.H, W=sc.W, use_rgb=True)[0]
        with torch.no_grad():
            p["opacity"].mul_(0.9)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            out.backward()
    R.check_overflow()


def test_the_rerun_after_an_overflow_does_not_walk_the_overflowed_lists(dev):
    """Found by tools/fuzz_dropin.py --shrink: a sync-free call overflows its instance capacity (NaN image), check_overflow() says so
    and raises the capacity — and the re-run the message asks for, made with the same tensor objects and precomputed colours while the
    failed output is still alive, was taken for the MASK pass of the failed call (same objects, same camera, directly after a full
    call) and re-walked its truncated lists: NaN again, even with sync=True. A call whose counters are known to have overflowed (or
    are read now, by a call that reads D back anyway) is never a geometry parent."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=1200, use_rgb=True, blend=False)

    def render(sync):
        from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
                                           scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform.float(),
                                           sh_degree=0, campos=cam.camera_center, prefiltered=False, debug=False)
        return GaussianRasterizer(rs, sync=sync)(means3D=t["xyz"], means2D=torch.zeros_like(t["xyz"], requires_grad=True), opacities=t["opacity"],
                                                 scales=t["scaling"], rotations=t["rotation"], colors_precomp=t["colour"], cov3D_precomp=None)[0]

    from guassianhand_amd.camera import Camera
    H, W = 110, 66
    cam = Camera.from_w2c(sc.w2c[0].to(dev), sc.K[0].to(dev), H, W)
    bg = torch.zeros(3, device=dev)
    t = {k: v.to(dev).clone().requires_grad_(True) for k, v in dict(xyz=sc.xyz, opacity=sc.opacity, scaling=sc.scaling, rotation=sc.rotation,
                                                                   colour=sc.shs.squeeze(1)).items()}
    ref = render(True).detach().clone()
    key = R.capacity_key(sc.P, 1, H, W)
    saved = R._capacity[key]
    try:
        R._capacity[key] = max(64, R.last_num_rendered() // 3)
        failed = render(False)                                   # sync-free: overflows, NaN image, nobody has looked yet
        assert bool(torch.isnan(failed).any())
        with pytest.raises(R.GhOverflowError):
            R.check_overflow()
        assert R._capacity[key] > R.last_num_rendered()
        again = render(True)                                     # `failed` is still alive: its lists must not be walked again
        assert torch.equal(again.detach(), ref)
        # the verdict not yet read: a call that reads D back itself asks before it shares
        R._capacity[key] = max(64, R.last_num_rendered() // 3)
        R.set_geometry_reuse(True)                               # (forget `again`: the next call is a full one, not its mask pass)
        failed2 = render(False)
        again2 = render(True)
        assert torch.equal(again2.detach(), ref) and bool(torch.isnan(failed2).any())
        with pytest.raises(R.GhOverflowError):
            R.check_overflow()
    finally:
        R._capacity[key] = max(saved, R._capacity.get(key, 0))
        try:
            R.check_overflow()
        except R.GhOverflowError:
            pass


def test_no_gaussians(dev):
    """P = 0. The C-ABI composites the background over nothing (T = 1 everywhere: App. A's formulas) and its backward produces empty
    gradients; the drop-in module returns what the published wrapper returns for an empty model — the zero image it allocated, the
    kernels skipped — and stays differentiable (empty gradients)."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from guassianhand_amd.camera import Camera
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=10)
    cams = sc.cams().to(dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    img, radii, ctx = R.raster_forward(cams, z(0, 3), z(0, 1), z(0, 3), z(0, 4), H=40, W=56, colors_precomp=z(0, 3))
    bg = cams[:, 37:40]
    assert img.shape == (2, 3, 40, 56) and radii.shape == (2, 0)
    assert torch.equal(img, bg[:, :, None, None].expand(2, 3, 40, 56))
    g = R.raster_backward(ctx, torch.randn(2, 3, 40, 56, device=dev), want_means2D=False)
    assert all(v.numel() == 0 for v in g.values())
    cam = Camera.from_w2c(sc.w2c[0].to(dev), sc.K[0].to(dev), 40, 56)
    rs = GaussianRasterizationSettings(image_height=40, image_width=56, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                       bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                       projmatrix=cam.full_proj_transform.float(), sh_degree=0, campos=cam.camera_center, prefiltered=False, debug=False)
    xyz = z(0, 3).requires_grad_(True)
    out, rad = GaussianRasterizer(rs)(means3D=xyz, means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4),
                                      cov3D_precomp=None)
    assert out.shape == (3, 40, 56) and float(out.detach().abs().max()) == 0.0 and rad.numel() == 0
    out.sum().backward()
    assert xyz.grad is not None and xyz.grad.shape == (0, 3)
    R.check_overflow()
