"""The HIP path against Oracle A (dense float64 autograd) IN THE REGIME THE BENCHMARK RUNS IN (VERDICT r3 'next' item 2).

tests/test_gpu_oracle_a.py is the independent leg at BASELINE configs[0]'s size (1k Gaussians, 128x128, f = 325). The hand scenes
— f = 1300 / 2600, 2 mm Gaussians stacked a thousand deep, 512x334 / 1024x1024, RGB + blend / SH3 + blend — were only ever
compared with Oracle B, whose per-Gaussian stage is the kernels' own prose contract typed a second time. A dense P x H x W
float64 evaluation of 98,562 Gaussians is out of reach, so the comparison is made on a WINDOW of the real render:

  1. the full scene is rendered by the HIP path (P = 98,562, full image);
  2. the Gaussians whose 3-sigma tile rect touches the window's tiles (+ one tile of margin) are selected — every Gaussian
     that can reach a pixel of the window; the HIP render of that SUBSET at the full image size must equal the full render
     inside the window BIT FOR BIT (so the subset stands for the real scene, nothing is re-tuned);
  3. Oracle A evaluates the subset on the window's pixels only (`pixel_window`: projection, rects and tile membership are
     those of the full image) in float64, backward by autograd, chunk-checkpointed;
  4. image and every gradient (dL/dimage non-zero inside the window only) are compared at the north star's tolerances with
     the float32-vs-float64 flip accounting of tests/test_gpu_oracle_a.py; the HIP gradients of the FULL scene under the same
     dL/dimage are checked against the subset's (zero outside the subset).
"""
import pytest
import torch

from tests.helpers import dimg_like, max_rel, rel_l2

pytestmark = pytest.mark.gpu

IMG_LINF = 1e-4
GRAD_RTOL = 1e-3
GRAD_L2 = 2e-5
FLIP_LINF = 1e-2
ROW_RTOL = 1e-4
# Windows are thousands of pixels under lists a thousand entries deep: ~1e7 (pixel, Gaussian) alpha tests per window at about
# one float32-vs-float64 threshold flip per million (measured in test_gpu_oracle_a.py) -> allow a proportionate count. (Measured
# against Oracle B on these windows: none above 1e-4 — a flip deep in a list is weighted by a small transmittance.)
FLIPS_PER_MILLION_TESTS = 3.0
# Under a list a thousand entries deep one alpha-vs-1/255 decision taken differently moves the PIXEL by alpha * T * c < 1e-4 (T is
# small there) — invisible to the image bar — but moves the gradient of a faint Gaussian, which is made of a handful of such
# terms, by tens of percent (seen: one Gaussian of 15,261, opacity 0.08, row error 19 % with an image error of 6.7e-5). A pixel
# whose list holds a decision within AMBIGUITY_EPS (relative) of its threshold — evaluated by Oracle A itself, in float64 —
# is therefore left out of the loss on both sides; the float32 rounding of alpha is ~1e-6 relative, so the margin is 20x that.
# The fraction of such pixels is printed and bounded.
AMBIGUITY_EPS = 2e-5
MAX_AMBIGUOUS_FRACTION = 0.02
MAX_CANCELLED = 4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    from tests.helpers import limit_torch_threads_to_the_cpu_share
    _lib.lib()
    limit_torch_threads_to_the_cpu_share()          # Oracle A runs on the host's CPUs: use the share this process owns
    return torch.device("cuda:0")


def _subset(sc, idx):
    import dataclasses
    per = {k: getattr(sc, k)[idx].contiguous() for k in ("xyz", "opacity", "rotation", "scaling", "shs", "color_b", "opacity_b")
           if getattr(sc, k) is not None}
    return dataclasses.replace(sc, **per)


def _hip(sc, dev, dimg, want_ctx=False):
    from guassianhand_amd.rasterizer import raster_backward, raster_forward, workspace_views
    s = sc.to(dev)
    kw = dict(colors_precomp=s.shs.squeeze(1)) if sc.use_rgb else dict(shs=s.shs, sh_degree=sc.sh_degree)
    bl = {k: getattr(s, k) for k in ("xyz_b", "opacity_b", "color_w", "color_b") if getattr(s, k) is not None}
    img, radii, ctx = raster_forward(sc.cams().to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
    rect = workspace_views(ctx)["rect"].long().cpu()
    grads = None
    if dimg is not None:
        grads = {k: v.cpu() for k, v in raster_backward(ctx, dimg.to(dev), want_means2D=False).items()}
    torch.cuda.synchronize()
    return img.cpu(), radii.cpu(), rect, grads


def _oracle_a_window(sc, win):
    from oracle import oracle_torch as OT
    d = torch.float64
    leaves = {n: getattr(sc, n).to(d).clone().requires_grad_(True) for n in ("xyz", "opacity", "scaling", "rotation", "shs")}
    bl = {k: getattr(sc, k).to(d).clone().requires_grad_(True) for k in ("color_w", "xyz_b", "color_b", "opacity_b") if getattr(sc, k) is not None}
    c = sc.cams()[0].to(d)
    means, opac, cols, sh = OT.blend_attributes(leaves["xyz"], leaves["opacity"], leaves["shs"], use_rgb=sc.use_rgb, **bl)
    kw = dict(colors_precomp=cols) if sc.use_rgb else dict(shs=sh, sh_degree=sc.sh_degree)
    img, _, amb = OT.rasterize_dense(means, opac, leaves["scaling"], leaves["rotation"], viewmatrix=c[:16].reshape(4, 4),
                                     projmatrix=c[16:32].reshape(4, 4), campos=c[32:35], tanfovx=float(c[35]), tanfovy=float(c[36]),
                                     bg=c[37:40], H=sc.H, W=sc.W, pixel_window=win, checkpoint_chunks=True, pixel_chunk=256,
                                     ambiguity_eps=AMBIGUITY_EPS, **kw)

    def backward(dwin):
        (img * dwin.to(d)).sum().backward()
        g = dict(means3D=leaves["xyz"].grad, opacities=leaves["opacity"].grad, scales=leaves["scaling"].grad, rotations=leaves["rotation"].grad)
        g["colors_precomp" if sc.use_rgb else "shs"] = leaves["shs"].grad
        g.update({k: v.grad for k, v in bl.items()})
        return g
    return img.detach(), amb, backward


# name: (scene config, view of the 2-view ring, window in tiles (tx0, ty0, n) — centre of the hands, where the lists are deepest)
CASES = {
    "two_hands-f1300-512x334-rgb-blend-64px": ("two_hands", 0, (8, 14, 4)),
    "two_hands-f1300-512x334-rgb-blend-silhouette-96px": ("two_hands", 0, (2, 13, 6)),
    "two_hands_hd-f2600-1024-sh3-blend-96px": ("two_hands_hd", 0, (29, 29, 6)),
}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("case", list(CASES))
def test_hand_scene_window_matches_the_dense_float64_autograd_oracle(dev, case):
    from guassianhand_amd.scenes import make_scene
    config, view, (tx0, ty0, n) = CASES[case]
    sc = make_scene(config, n_views=2)
    sc.w2c, sc.K = sc.w2c[view:view + 1].contiguous(), sc.K[view:view + 1].contiguous()
    sc.xyz_b = torch.tensor([0.0015, -0.001, 0.002])                      # a non-zero position bias (the scenes ship zeros)
    assert sc.P == 98562
    H, W = sc.H, sc.W
    win = (tx0 * 16, ty0 * 16, min(W, (tx0 + n) * 16), min(H, (ty0 + n) * 16))
    x0, y0, x1, y1 = win

    dimg = torch.zeros(1, 3, H, W)
    dwin = dimg_like(1, y1 - y0, x1 - x0, seed=33)[0]
    dimg[0, :, y0:y1, x0:x1] = dwin
    img_full, radii_full, rect, g_full = _hip(sc, dev, dimg)
    # every Gaussian whose 3-sigma rect touches the window's tiles, one tile of margin on every side
    r = rect[:sc.P]
    minx, miny, maxx, maxy = r & 255, (r >> 8) & 255, (r >> 16) & 255, (r >> 24) & 255
    sel = (maxx > tx0 - 1) & (minx < tx0 + n + 1) & (maxy > ty0 - 1) & (miny < ty0 + n + 1) & (radii_full[0] > 0)
    idx = torch.nonzero(sel).reshape(-1)
    sub = _subset(sc, idx)
    print(f"{case}: window {win}, {idx.numel()} of {sc.P} Gaussians reach it")
    assert 1000 < idx.numel() < 40000

    # (2) the subset IS the scene inside the window: bit for bit, forward; gradients of the full scene == the subset's
    img_sub, _, _, g_sub = _hip(sub, dev, dimg)
    assert torch.equal(img_sub[0, :, y0:y1, x0:x1], img_full[0, :, y0:y1, x0:x1])
    for k, v in g_full.items():
        if v.dim() >= 1 and v.shape[0] == sc.P:
            rest = torch.ones(sc.P, dtype=torch.bool)
            rest[idx] = False
            assert float(v[rest].abs().max()) == 0.0, k                   # nothing outside the subset can see the window
            assert rel_l2(v[idx], g_sub[k]) <= 1e-6, (k, rel_l2(v[idx], g_sub[k]))
        else:
            assert rel_l2(v, g_sub[k]) <= 1e-6, (k, rel_l2(v, g_sub[k]))

    # (3) + (4) Oracle A, float64, window pixels only. (Flush-to-zero for the CPU evaluation: the running transmittance under a
    # thousand-deep list underflows into float64 denormals, which cost 50x the time and contribute exactly nothing.)
    ftz = torch.set_flush_denormal(True)
    img_a, ambiguous, backward_a = _oracle_a_window(sub, win)
    err = (img_sub[0, :, y0:y1, x0:x1].double() - img_a).abs().amax(dim=0)       # the image bar holds on EVERY pixel but counted flips
    flipped = err > IMG_LINF
    n_flip = int(flipped.sum())
    frac_amb = float(ambiguous.float().mean())
    print(f"{case}: {int(ambiguous.sum())} of {ambiguous.numel()} pixels ({100 * frac_amb:.2f} %) hold a decision within {AMBIGUITY_EPS:g} of "
          f"its threshold: left out of the gradient comparison")
    assert frac_amb <= MAX_AMBIGUOUS_FRACTION
    tests_m = idx.numel() * (y1 - y0) * (x1 - x0) / 1e6 * 0.05            # ~5 % of the (pixel, Gaussian) pairs of a window share a tile rect
    allowed = max(2, int(FLIPS_PER_MILLION_TESTS * tests_m))
    print(f"{case}: image L_inf {float(err[~flipped].max()):.3g} over {int((~flipped).sum())} pixels; {n_flip} pixel(s) with a float32-vs-"
          f"float64 threshold decision (allowed {allowed}), L_inf {float(err[flipped].max()) if n_flip else 0.0:.3g}")
    assert n_flip <= allowed and (n_flip == 0 or float(err[flipped].max()) <= FLIP_LINF)
    dwin2 = dwin * (~(flipped | ambiguous))[None].float()
    g_a = backward_a(dwin2)
    torch.set_flush_denormal(False)
    dimg2 = torch.zeros(1, 3, H, W)
    dimg2[0, :, y0:y1, x0:x1] = dwin2
    _, _, _, grads = _hip(sub, dev, dimg2)
    assert set(g_a) <= set(grads), (sorted(g_a), sorted(grads))
    for k, ga in g_a.items():
        gh = grads[k].double().reshape(ga.shape)
        assert bool(torch.isfinite(gh).all()), k
        print(f"{case}: {k}: rel-L2 {rel_l2(gh, ga):.3g}, max-rel {max_rel(gh, ga):.3g}")
        assert rel_l2(gh, ga) <= GRAD_L2, (k, rel_l2(gh, ga))
        if max_rel(gh, ga) > GRAD_RTOL:
            a2 = gh.reshape(ga.shape[0], -1) if ga.dim() > 1 else gh.reshape(1, -1)
            b2 = ga.reshape(ga.shape[0], -1) if ga.dim() > 1 else ga.reshape(1, -1)
            el = (a2 - b2).abs() / (b2.abs() + 1e-3 * b2.abs().max())
            bad = torch.nonzero(el > GRAD_RTOL)
            row = (a2 - b2).norm(dim=1) / (b2.norm(dim=1) + 1e-30)
            print(f"{case}: {k}: {bad.shape[0]} element(s) above the element-wise bar (max {float(el.max()):.3g}), "
                  f"row-relative error there {float(row[bad[:, 0]].max()):.3g}")
            assert bad.shape[0] <= MAX_CANCELLED and float(row[bad[:, 0]].max()) <= ROW_RTOL, (k, max_rel(gh, ga))
