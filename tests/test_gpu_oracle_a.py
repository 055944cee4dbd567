"""The HIP path against Oracle A directly (VERDICT r2 'next' item 4).

Every other GPU parity test checks the kernels against Oracle B (`oracle/gh_oracle.c`), whose per-Gaussian stage is the same
prose contract as the kernels' typed a second time — bit-exactness between the two proves consistent typing, not independent
correctness. Oracle A (`oracle/oracle_torch.py`) is a different program: a dense pixel x Gaussian evaluation written from the
behavioural spec, no tiles, no sort keys, no hand-written chain rule — its backward is PyTorch autograd —
and it runs here in FLOAT64. Agreement of the fp32 HIP path with it, at BASELINE configs[0]'s size, with the north star's own
tolerances (RGB L_inf <= 1e-4, gradient rtol <= 1e-3), is the independent leg of the GPU record.
"""
import pytest
import torch

from tests.helpers import dimg_like, max_rel, rel_l2

pytestmark = pytest.mark.gpu

IMG_LINF = 1e-4
GRAD_RTOL = 1e-3
GRAD_L2 = 2e-5
# A float64 evaluation and a float32 one can take a discrete decision of App. A.3 differently when a value sits within float32
# rounding of its threshold (alpha against 1/255: about one (pixel, Gaussian) test in a million on these scenes). Such a pixel
# differs by one skipped / blended entry, alpha ~ 1/255: it is COUNTED and BOUNDED here, and left out of the gradient
# comparison (its dL/dpixel is zeroed on both sides) — everything else has to meet the north star's tolerances.
MAX_FLIPPED_PIXELS = 2
FLIP_LINF = 1e-2
# Gradients: element-wise |a - b| <= GRAD_RTOL * (|b| + 1e-3 max|b|) (tests.helpers.max_rel). The float32 chain rule loses
# absolute accuracy where one COMPONENT of a Gaussian's gradient vector is the small residual of cancelling terms a thousand
# times larger (its siblings): such an element may miss the element-wise bar against a float64 oracle while the vector is
# accurate to 1e-5. At most MAX_CANCELLED elements per tensor may do so, each within ROW_RTOL of its row's norm — counted
# and printed, never silently widened. (Seen: one element of one case, 1.4e-3; Oracle B, also float32, shows the same value.)
MAX_CANCELLED = 2
ROW_RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    from tests.helpers import limit_torch_threads_to_the_cpu_share
    _lib.lib()
    limit_torch_threads_to_the_cpu_share()          # Oracle A runs on the host's CPUs: use the share this process owns
    return torch.device("cuda:0")


def _oracle_a(sc, blend):
    """float64 dense render of every view, with its autograd graph: (images, backward(dimg) -> gradients of sum(img * dimg)
    w.r.t. every leaf incl. the blend parameters)."""
    from oracle import oracle_torch as OT
    d = torch.float64
    leaves = {n: getattr(sc, n).to(d).clone().requires_grad_(True) for n in ("xyz", "opacity", "scaling", "rotation", "shs")}
    bl = {k: v.to(d).clone().requires_grad_(True) for k, v in blend.items()}
    cams = sc.cams().to(d)
    imgs = []
    for v in range(cams.shape[0]):
        c = cams[v]
        means, opac, cols, sh = OT.blend_attributes(leaves["xyz"], leaves["opacity"], leaves["shs"], use_rgb=sc.use_rgb, **bl)
        kw = dict(colors_precomp=cols) if sc.use_rgb else dict(shs=sh, sh_degree=sc.sh_degree)
        img, _ = OT.rasterize_dense(means, opac, leaves["scaling"], leaves["rotation"], viewmatrix=c[:16].reshape(4, 4),
                                    projmatrix=c[16:32].reshape(4, 4), campos=c[32:35], tanfovx=float(c[35]), tanfovy=float(c[36]),
                                    bg=c[37:40], H=sc.H, W=sc.W, **kw)
        imgs.append(img)
    imgs = torch.stack(imgs)

    def backward(dimg):
        (imgs * dimg.to(d)).sum().backward()
        grads = dict(means3D=leaves["xyz"].grad, opacities=leaves["opacity"].grad, scales=leaves["scaling"].grad,
                     rotations=leaves["rotation"].grad)
        grads["colors_precomp" if sc.use_rgb else "shs"] = leaves["shs"].grad
        grads.update({k: v.grad for k, v in bl.items()})
        return grads
    return imgs.detach(), backward


CASES = {
    # name: (use_rgb, which blend parameters are given, per-Gaussian color_w)
    "rgb-plain": (True, (), False),
    "rgb-blend-all": (True, ("color_w", "color_b", "opacity_b", "xyz_b"), False),
    "rgb-blend-w-per-gaussian": (True, ("color_w", "color_b", "opacity_b", "xyz_b"), True),
    "rgb-w-only": (True, ("color_w",), False),
    "rgb-b-and-opacity-only": (True, ("color_b", "opacity_b"), False),
    "sh3-plain": (False, (), False),
    "sh3-blend-all": (False, ("color_w", "color_b", "opacity_b", "xyz_b"), False),      # incl. the double multiply of :334
    "sh3-blend-w-per-gaussian": (False, ("color_w", "color_b", "opacity_b", "xyz_b"), True),
    "sh3-w-only": (False, ("color_w",), False),
}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", list(CASES))
def test_hip_path_matches_the_dense_float64_autograd_oracle(dev, case):
    from guassianhand_amd.rasterizer import raster_backward, raster_forward
    from guassianhand_amd.scenes import make_scene
    use_rgb, given, wpg = CASES[case]
    sc = make_scene("random1k", n_views=2, use_rgb=use_rgb, blend=True)            # BASELINE configs[0]: 1k Gaussians, 128x128
    g = torch.Generator().manual_seed(13)
    sc.xyz_b = 0.004 * torch.randn(3, generator=g)
    if wpg:
        sc.color_w = 1 + 0.05 * torch.randn(sc.P, 48, generator=g)                 # the edit renderer's (P,48) weights
    blend = {k: getattr(sc, k) for k in given}
    dimg = dimg_like(2, sc.H, sc.W, seed=21)
    img_a, backward_a = _oracle_a(sc, blend)

    s = sc.to(dev)
    kw = dict(colors_precomp=s.shs.squeeze(1)) if use_rgb else dict(shs=s.shs, sh_degree=sc.sh_degree)
    img, _, ctx = raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W,
                                 **{k: getattr(s, k) for k in given}, **kw)
    err = (img.double().cpu() - img_a).abs().amax(dim=1)               # (views, H, W): worst channel of every pixel
    flipped = err > IMG_LINF
    n_flip = int(flipped.sum())
    print(f"{case}: image L_inf {float(err[~flipped].max()):.3g} over {int((~flipped).sum())} pixels; {n_flip} pixel(s) with a "
          f"float32-vs-float64 threshold decision, L_inf {float(err[flipped].max()) if n_flip else 0.0:.3g}")
    assert n_flip <= MAX_FLIPPED_PIXELS and (n_flip == 0 or float(err[flipped].max()) <= FLIP_LINF)
    dimg = dimg * (~flipped)[:, None].float()
    g_a = backward_a(dimg)
    grads = raster_backward(ctx, dimg.to(dev), want_means2D=False)
    torch.cuda.synchronize()
    assert set(g_a) <= set(grads), (sorted(g_a), sorted(grads))
    for k, ga in g_a.items():
        gh = grads[k].double().cpu().reshape(ga.shape)
        assert bool(torch.isfinite(gh).all()), k
        assert rel_l2(gh, ga) <= GRAD_L2, (k, rel_l2(gh, ga))
        if max_rel(gh, ga) > GRAD_RTOL:
            a2, b2 = gh.reshape(ga.shape[0], -1) if ga.dim() > 1 else gh.reshape(1, -1), ga.reshape(ga.shape[0], -1) if ga.dim() > 1 else ga.reshape(1, -1)
            el = (a2 - b2).abs() / (b2.abs() + 1e-3 * b2.abs().max())
            bad = torch.nonzero(el > GRAD_RTOL)
            row = (a2 - b2).norm(dim=1) / (b2.norm(dim=1) + 1e-30)
            print(f"{case}: {k}: {bad.shape[0]} element(s) above the element-wise bar (max {float(el.max()):.3g}), "
                  f"row-relative error there {float(row[bad[:, 0]].max()):.3g}")
            assert bad.shape[0] <= MAX_CANCELLED and float(row[bad[:, 0]].max()) <= ROW_RTOL, (k, max_rel(gh, ga))
