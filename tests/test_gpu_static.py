"""Static geometry (VERDICT r2 'next' item 6): tile lists built once with GH_FLAG_STATIC_LISTS and re-used by
gh_forward_refresh / gh_backward_refresh while only opacities and colours move — the one-shot fit's step
(infer_one_shot.py:489-524: same Gaussians, same cameras, trained colour / opacity biases, renderer_one_shot.py:306-334).

Contract checked here: forward images of a refresh call == a full call with the same inputs, bit for bit; gradients agree to
rounding (another partition of the lists into depth segments); an opacity above the lists' bound can never produce a
plausible image; any change of a geometry input falls back to the full path by itself.
"""
import pytest
import torch

from tests.helpers import dimg_like, max_rel, rel_l2, scene_kwargs, tiny_fit_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _moved(s, seed):
    """Other opacities / colours for the same geometry: what a few Adam steps on color_w / color_b / opacity_b do."""
    g = torch.Generator().manual_seed(seed)
    dev = s.xyz.device
    r = lambda t, a: t + (a * torch.randn(t.shape, generator=g)).to(dev)
    return dict(opacity_b=r(s.opacity_b, 0.05), color_w=r(s.color_w, 0.05), color_b=r(s.color_b, 0.05), xyz_b=s.xyz_b)


def _grads_close(a, b, names=None):
    for k in (names or b):
        assert rel_l2(a[k], b[k]) <= 1e-5, (k, rel_l2(a[k], b[k]))
        assert max_rel(a[k], b[k]) <= 1e-3, (k, max_rel(a[k], b[k]))


@pytest.mark.parametrize("use_rgb,nv,alpha", [(True, 3, True), (False, 2, False), (True, 1, False)])
def test_refresh_call_equals_a_full_call_with_the_same_inputs(dev, use_rgb, nv, alpha):
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=nv, P=3000, use_rgb=use_rgb, blend=True)
    s = sc.to(dev)
    kw, bl0 = scene_kwargs(s)
    cams = s.cams().contiguous()
    geo = (cams, s.xyz, s.opacity, s.scaling, s.rotation)
    d = dimg_like(nv, sc.H, sc.W).to(dev)
    da = torch.randn(nv, sc.H, sc.W, generator=torch.Generator().manual_seed(2)).to(dev) if alpha else None

    # the build call is a plain call with longer lists: same image, same radii, gradients to rounding
    img_p, radii_p, ctx_p = R.raster_forward(*geo, H=sc.H, W=sc.W, return_alpha=alpha, **kw, **bl0)
    D_plain = R.last_num_rendered()
    g_p = R.raster_backward(ctx_p, d, want_means2D=False, dL_dalpha=da)
    img_s, radii_s, ctx_s = R.raster_forward(*geo, H=sc.H, W=sc.W, return_alpha=alpha, static_lists=True, **kw, **bl0)
    D_static = R.last_num_rendered()
    g_s = R.raster_backward(ctx_s, d, want_means2D=False, dL_dalpha=da)
    assert torch.equal(img_s, img_p) and torch.equal(radii_s, radii_p) and D_static >= D_plain > 0
    if alpha:
        assert torch.equal(ctx_s.alpha, ctx_p.alpha)
    _grads_close(g_s, g_p)

    # later steps: the same geometry with moved opacities / colours
    for step in range(3):
        bl = _moved(s, 10 + step)
        img_f, _, ctx_f = R.raster_forward(*geo, H=sc.H, W=sc.W, return_alpha=alpha, **kw, **bl)
        g_f = R.raster_backward(ctx_f, d, want_means2D=False, dL_dalpha=da)
        img_r, radii_r, ctx_r = R.raster_forward(*geo, H=sc.H, W=sc.W, return_alpha=alpha, refresh_of=ctx_s, **kw, **bl)
        assert torch.equal(img_r, img_f), float((img_r - img_f).abs().max())
        assert torch.equal(radii_r, radii_p)
        if alpha:
            assert torch.equal(ctx_r.alpha, ctx_f.alpha)
        g_r = R.raster_backward(ctx_r, d, want_means2D=False, dL_dalpha=da)
        assert set(g_r) == set(g_f)
        _grads_close(g_r, g_f)
        # only what the fit trains: the chain rule through the projection is skipped, the sums are the same sums
        want = {"opacity_b", "color_w", "color_b"}
        _, _, ctx_r2 = R.raster_forward(*geo, H=sc.H, W=sc.W, return_alpha=alpha, refresh_of=ctx_s, **kw, **bl)
        g_l = R.raster_backward(ctx_r2, d, want_means2D=False, dL_dalpha=da, want=want)
        assert set(g_l) == want
        for k in want:
            assert torch.equal(g_l[k], g_r[k]), k
    R.check_overflow()


def test_an_opacity_above_the_bound_poisons_the_refresh_and_the_cache_rebuilds(dev):
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=2500, use_rgb=True, blend=True)
    s = sc.to(dev)
    kw, bl = scene_kwargs(s)
    cams = s.cams().contiguous()
    geo = (cams, s.xyz, s.opacity, s.scaling, s.rotation)
    cache = R.GeometryCache()
    img0, _, c0 = R.cached_raster_forward(cache, *geo, H=sc.H, W=sc.W, **kw, **bl)
    assert (cache.builds, cache.hits) == (1, 0) and cache.ctx is c0
    img1, _, c1 = R.cached_raster_forward(cache, *geo, H=sc.H, W=sc.W, **kw, **bl)
    assert (cache.builds, cache.hits) == (1, 1) and c1.parent is c0 and torch.equal(img1, img0)
    # one visible Gaussian's opacity goes far above 1: its alpha >= 1/255 ellipse grows beyond the tiles listed for it
    vis = int(torch.nonzero(c0.radii[0] > 0)[0])
    ob = bl["opacity_b"].clone()
    ob[vis] = 40.0
    bl_hi = dict(bl, opacity_b=ob)
    ref, _, _ = R.raster_forward(*geo, H=sc.H, W=sc.W, **kw, **bl_hi)
    # sync-free: NaN image, the error at the check, the caches cleared
    img_bad, _, _ = R.raster_forward(*geo, H=sc.H, W=sc.W, refresh_of=c0, sync=False, **kw, **bl_hi)
    assert torch.isnan(img_bad).all()
    with pytest.raises(R.GhStaleGeometryError):
        R.check_overflow()
    assert cache.ctx is None
    # through the cache with a read-back (the default of the fit's first steps): rebuilt transparently, result of a full call
    cache.store((cams, s.xyz, s.scaling, s.rotation, bl["xyz_b"]), cache_vals(cams, s, bl, sc), c0)
    img2, _, c2 = R.cached_raster_forward(cache, *geo, H=sc.H, W=sc.W, **kw, **bl_hi)
    assert torch.equal(img2, ref) and c2.parent is None and cache.ctx is c2
    # ... and the rebuilt lists carry the new bound: the next step refreshes again
    img3, _, c3 = R.cached_raster_forward(cache, *geo, H=sc.H, W=sc.W, **kw, **bl_hi)
    assert c3.parent is c2 and torch.equal(img3, ref)
    R.check_overflow()


def cache_vals(cams, s, bl, sc):
    objs = (cams, s.xyz, s.scaling, s.rotation, bl["xyz_b"])
    return tuple(None if o is None else (o._version, tuple(o.shape), o.device) for o in objs) + (sc.H, sc.W, 1.0, False)


def test_any_change_of_the_geometry_is_a_cache_miss(dev):
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=2000, use_rgb=True, blend=True)
    s = sc.to(dev)
    kw, bl = scene_kwargs(s)
    cams = s.cams().contiguous()
    xyz = s.xyz.clone()
    cache = R.GeometryCache()
    call = lambda c=cams, x=xyz, **o: R.cached_raster_forward(cache, c, x, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W,
                                                              **kw, **dict(bl, **o))
    call(); call()
    assert (cache.builds, cache.hits) == (1, 1)
    call(opacity_b=bl["opacity_b"] * 0.5)                   # opacities / colours may move: still a hit
    assert (cache.builds, cache.hits) == (1, 2)
    xyz.add_(0.001)                                         # an in-place update bumps _version: rebuilt, and rendered where it IS
    img, _, ctx = call()
    assert (cache.builds, cache.hits) == (2, 2) and ctx.parent is None
    ref, _, _ = R.raster_forward(cams, xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
    assert torch.equal(img, ref)
    cams2 = cams.clone()
    call(c=cams2)                                           # another camera tensor
    assert cache.builds == 3
    call(c=cams2)
    assert (cache.builds, cache.hits) == (3, 3)
    call(c=cams2, xyz_b=bl["xyz_b"] + 0.01)                 # another xyz_b object
    assert cache.builds == 4
    cache.clear()
    call(c=cams2)
    assert cache.builds == 5
    R.check_overflow()


@pytest.mark.parametrize("use_rgb", [True, False])
def test_fit_with_static_geometry_takes_the_steps_of_the_full_path(dev, use_rgb):
    from guassianhand_amd import fit as F
    from guassianhand_amd import rasterizer as R
    pb = tiny_fit_problem(P=600, n_views=4, hw=(64, 64), device=dev)
    g = torch.Generator().manual_seed(4)
    gt_rgb = torch.rand(4, 64, 64, 3, generator=g).to(dev)
    gt_mask = (torch.rand(4, 64, 64, generator=g) > 0.5).float().to(dev)
    args = (pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
    gs = pb["gs"]
    if not use_rgb:
        from guassianhand_amd.renderer import GaussianModel
        shs = torch.cat([gs.shs, 0.1 * torch.randn(gs.shs.shape[0], 15, 3, generator=g).to(dev)], 1)
        gs = GaussianModel(gs.xyz, gs.opacity, gs.rotation, gs.scaling, shs)
    mk = lambda static: F.OneShotFit(gs, pb["uv"], map_hw=pb["map_hw"], use_rgb=use_rgb, static_geometry=static)
    a, b = mk(False), mk(True)
    assert a._geom_cache is None and b._geom_cache is not None
    n = 8
    la = [float(a.step(*args, sync=(i == 0))) for i in range(n)]
    lb = [float(b.step(*args, sync=(i == 0))) for i in range(n)]
    R.check_overflow()
    assert (b._geom_cache.builds, b._geom_cache.hits) == (1, n - 1)
    assert lb == pytest.approx(la, rel=2e-6)
    for k in a._adam:
        assert torch.allclose(a._adam[k].param, b._adam[k].param, rtol=1e-4, atol=2e-6), k
    # other cameras (a new tensor): rebuilt by itself
    w2c2 = pb["w2c"].clone()
    b.step(w2c2, *args[1:], sync=True)
    assert b._geom_cache.builds == 2
    b.invalidate_geometry()
    b.step(w2c2, *args[1:], sync=True)
    assert b._geom_cache.builds == 3
    R.check_overflow()


def test_gaussians_that_move_between_steps_take_the_full_path(dev):
    """VERDICT r2 missing 4: a fit whose network-side trainables move the Gaussians (map_bias, infer_one_shot.py:340-343) calls
    update_gaussians(): blend maps and Adam state carry over, the static lists are rebuilt, and the step equals the step of a
    fit that was constructed at the new geometry with the same maps."""
    from guassianhand_amd import fit as F
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.renderer import GaussianModel
    pb = tiny_fit_problem(P=600, n_views=4, hw=(64, 64), device=dev)
    g = torch.Generator().manual_seed(6)
    gt_rgb = torch.rand(4, 64, 64, 3, generator=g).to(dev)
    gt_mask = (torch.rand(4, 64, 64, generator=g) > 0.5).float().to(dev)
    args = (pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
    gs0 = pb["gs"]
    gs1 = GaussianModel(gs0.xyz + 0.002 * torch.randn(gs0.xyz.shape, generator=g).to(dev), gs0.opacity, gs0.rotation, gs0.scaling, gs0.shs)
    a = F.OneShotFit(gs0, pb["uv"], map_hw=pb["map_hw"])
    for i in range(3):
        a.step(*args, sync=True)
    b = F.OneShotFit(gs1, pb["uv"], map_hw=pb["map_hw"])                 # the same maps / moments, constructed at the new geometry
    for k in a._adam:
        for n_ in ("param", "exp_avg", "exp_avg_sq", "step_state"):
            getattr(b._adam[k], n_).copy_(getattr(a._adam[k], n_))
        b._adam[k].t = a._adam[k].t
    builds = a._geom_cache.builds
    a.update_gaussians(gs1)
    la, lb = float(a.step(*args, sync=True)), float(b.step(*args, sync=True))
    assert a._geom_cache.builds == builds + 1
    assert la == lb
    for k in a._adam:
        assert torch.equal(a._adam[k].param, b._adam[k].param), k
    with pytest.raises(ValueError):
        a.update_gaussians(GaussianModel(*[t[:10] for t in gs1]))
    R.check_overflow()
