"""cov3D_precomp of the published module API (the reference never passes it, renderer_one_shot.py:313, :346): the HIP path against
the C oracle, through the C-ABI (`raster_forward`) and through the drop-in module with autograd."""
import pytest
import torch

from tests.helpers import dimg_like, max_rel, rel_l2
from tests.test_cov3d_cpu import _sigma6

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("use_rgb,nv,blend", [(True, 1, False), (True, 3, True), (False, 2, True)])
def test_cov3d_precomp_matches_the_oracle(dev, use_rgb, nv, blend):
    from guassianhand_amd.rasterizer import raster_backward, raster_forward
    from guassianhand_amd.scenes import make_scene
    from oracle.oracle_c import OracleRender
    from tests.helpers import scene_kwargs
    sc = make_scene("random1k", n_views=nv, P=1200, use_rgb=use_rgb, blend=blend)
    cov = _sigma6(sc).float()
    kw, bl = scene_kwargs(sc)
    o = OracleRender(sc.cams(), sc.xyz, sc.opacity, None, None, H=sc.H, W=sc.W, cov3D_precomp=cov, **kw, **bl)
    s = sc.to(dev)
    kwd, bld = scene_kwargs(s)
    img, radii, ctx = raster_forward(s.cams(), s.xyz, s.opacity, None, None, H=sc.H, W=sc.W, cov3D_precomp=cov.to(dev), **kwd, **bld)
    assert torch.equal(radii.cpu(), o.radii)
    assert torch.equal(img.cpu(), o.image), "forward is expected to be bit-exact under the arithmetic contract"
    dimg = dimg_like(nv, sc.H, sc.W)
    g = raster_backward(ctx, dimg.to(dev))
    og = o.backward(dimg)
    assert "cov3D_precomp" in g and "scales" not in g and "rotations" not in g
    for k, b in og.items():
        if k == "means2D":
            continue
        a = g[k].cpu().reshape(b.shape)
        assert bool(torch.isfinite(a).all()), k
        assert rel_l2(a, b) <= 1e-5 and max_rel(a, b) <= 1e-3, (k, rel_l2(a, b), max_rel(a, b))
    o.close()


def test_cov3d_precomp_through_the_drop_in_module(dev):
    """The published keyword protocol with cov3D_precomp instead of scales / rotations: same validation messages, a (3,H,W) image,
    and a gradient for the covariance tensor; the picture equals the scales + rotations render of the same Gaussians to the
    float32 rounding of a covariance computed outside."""
    from guassianhand_amd.camera import Camera
    from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from guassianhand_amd.scenes import make_scene
    import math
    sc = make_scene("random1k", n_views=1, P=700)
    s = sc.to(dev)
    cam = Camera.from_w2c(s.w2c[0], s.K[0], sc.H, sc.W)
    rs = GaussianRasterizationSettings(image_height=sc.H, image_width=sc.W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                       projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center, prefiltered=False, debug=False)
    R = GaussianRasterizer(raster_settings=rs)
    cov = _sigma6(sc).float().to(dev).requires_grad_(True)
    xyz = s.xyz.clone().requires_grad_(True)
    cols = s.shs.squeeze(1)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        R(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=s.opacity, colors_precomp=cols, scales=s.scaling, rotations=s.rotation, cov3D_precomp=cov)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        R(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=s.opacity, colors_precomp=cols)
    img, radii = R(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=s.opacity, colors_precomp=cols, cov3D_precomp=cov)
    ref, _ = R(means3D=s.xyz, means2D=torch.zeros_like(xyz), opacities=s.opacity, colors_precomp=cols, scales=s.scaling, rotations=s.rotation)
    assert img.shape == (3, sc.H, sc.W) and radii.shape == (sc.P,)
    assert (img - ref).abs().max().item() <= 2e-5
    (img * dimg_like(1, sc.H, sc.W)[0].to(dev)).sum().backward()
    assert cov.grad is not None and cov.grad.shape == (sc.P, 6) and bool(torch.isfinite(cov.grad).all()) and float(cov.grad.abs().max()) > 0
    assert xyz.grad is not None and float(xyz.grad.abs().max()) > 0
