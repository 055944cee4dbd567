"""GH_FLAG_SPLIT_STREAMS: the views rendered as two halves on two HIP streams inside the library. Every output — images,
alpha, radii, all gradients, the instance count — must be the unsplit call's bit for bit (the halves are independent problems
whose per-(view, Gaussian) results land where the unsplit call puts them, and the chain rule runs once over all views)."""
import pytest
import torch

from tests.helpers import dimg_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _both(s, dev, n_views, *, shs_mode=False, alpha=False, per_view=False, views=None):
    from guassianhand_amd import rasterizer as R
    cams = s.cams() if views is None else s.cams()[views].contiguous()
    kw = dict(shs=s.shs, sh_degree=s.sh_degree) if shs_mode else dict(colors_precomp=s.shs.reshape(s.shs.shape[0], 3))
    d = dimg_like(n_views, s.H, s.W).to(dev)
    da = torch.rand(n_views, s.H, s.W, generator=torch.Generator().manual_seed(5)).to(dev) - 0.5 if alpha else None
    out = []
    for split in (False, True):
        img, radii, ctx = R.raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, xyz_b=s.xyz_b,
                                           opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b, sync=True,
                                           return_alpha=alpha, per_view_gaussians=per_view, split_streams=split, **kw)
        D = R.last_num_rendered()
        g = R.raster_backward(ctx, d, dL_dalpha=da)
        out.append((img, radii, ctx.alpha, g, D, bool(ctx.dims.flags & 8)))
    (i0, r0, a0, g0, D0, f0), (i1, r1, a1, g1, D1, f1) = out
    assert not f0 and f1
    assert torch.equal(i0, i1) and torch.equal(r0, r1) and D0 == D1
    if alpha:
        assert torch.equal(a0, a1)
    assert g0.keys() == g1.keys()
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    return i0


@pytest.mark.parametrize("n_views", [2, 3, 8])
def test_split_is_bit_identical_rgb_blend(dev, n_views):
    from guassianhand_amd.scenes import make_scene
    s = make_scene("two_hands", n_views=n_views, P=20000).to(dev)
    img = _both(s, dev, n_views)
    assert float(img.abs().sum()) > 0


def test_split_is_bit_identical_sh_and_alpha(dev):
    from guassianhand_amd.scenes import make_scene
    s = make_scene("two_hands_hd", n_views=4, P=6000).to(dev)
    _both(s, dev, 4, shs_mode=True, alpha=True)


def test_split_pose_batch(dev):
    import dataclasses
    from guassianhand_amd.scenes import SEED, make_scene
    B = 5
    poses = [make_scene("two_hands", n_views=B, P=4000, seed=SEED + 17 * b) for b in range(B)]
    cat = lambda k: None if getattr(poses[0], k) is None else torch.cat([getattr(p_, k) for p_ in poses])
    s = dataclasses.replace(poses[0], **{k: cat(k) for k in ("xyz", "opacity", "rotation", "scaling", "shs", "color_b", "opacity_b")}).to(dev)
    _both(s, dev, B, per_view=True)


def test_split_overflow_reports_needed_capacity(dev):
    """One half over its share: NaN image, GhOverflowError with a capacity that then suffices."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    s = make_scene("two_hands", n_views=4, P=20000).to(dev)
    cols = s.shs.reshape(s.shs.shape[0], 3)
    img, _, ctx = R.raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=True,
                                   split_streams=True)
    D = R.last_num_rendered()
    with pytest.raises(R.GhOverflowError):
        R.raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=True,
                         split_streams=True, max_instances=D // 2)
    # sync-free: NaN image from the device-side guard, the error at the check
    R.check_overflow()
    img2, _, _ = R.raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=False,
                                  split_streams=True, max_instances=D // 2)
    assert torch.isnan(img2).any()
    with pytest.raises(R.GhOverflowError):
        R.check_overflow()
    key = R.capacity_key(s.P, 4, s.H, s.W, True)
    need = R._capacity[key]
    img3, _, _ = R.raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=True,
                                  split_streams=True, max_instances=need)
    assert torch.equal(img3, img)


def test_split_call_that_fits_is_not_reported_as_overflowed(dev):
    """`reserved[0]` of a split call is the capacity that WOULD give each half a comfortable share (padded);
    it may exceed max_instances although neither half overflowed. Only GhCounters.overflow decides."""
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    s = make_scene("two_hands", n_views=4, P=20000).to(dev)
    cols = s.shs.reshape(s.shs.shape[0], 3)
    cams = s.cams()
    d_half = []
    for v0 in (0, 2):
        R.raster_forward(cams[v0:v0 + 2], s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=True)
        d_half.append(R.last_num_rendered())
    cap = 2 * max(d_half) + 128                       # each half's share (cap / 2, rounded down to 64) holds its instances
    ref, _, _ = R.raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=True)
    img, _, ctx = R.raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=True,
                                   split_streams=True, max_instances=cap)
    c = R.workspace_counters(ctx)
    assert c[1] & 15 == 0 and c[2] > cap              # fits (no error bit), yet the comfortable capacity is larger than the one given
    assert torch.equal(img, ref)
    R.check_overflow()
    img2, _, _ = R.raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols, sync=False,
                                  split_streams=True, max_instances=cap)
    R.check_overflow()                                # no error, no NaN
    assert torch.equal(img2, ref)


def test_split_inside_captured_graph(dev):
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    s = make_scene("two_hands", n_views=4, P=20000).to(dev)
    cols = s.shs.reshape(s.shs.shape[0], 3)
    d = dimg_like(4, s.H, s.W).to(dev)

    cams = s.cams().contiguous()

    def step():
        img, _, ctx = R.raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=cols,
                                       sync=False, split_streams=True)
        return img, R.raster_backward(ctx, d)
    img_ref, g_ref = step()
    R.check_overflow()
    R.set_graph_mode(True)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            img, g = step()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        R.check_overflow()
    finally:
        R.set_graph_mode(False)
    assert torch.equal(img, img_ref)
    for k in g:
        assert torch.equal(g[k], g_ref[k]), k
    # the library's side stream is back to plain (eager) use after the capture
    img_e, g_e = step()
    torch.cuda.synchronize()
    R.check_overflow()
    assert torch.equal(img_e, img_ref) and all(torch.equal(g_e[k], g_ref[k]) for k in g_ref)
