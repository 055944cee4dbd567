"""Soak test of the speculative occlusion bound: random adversarial scenes (dense stacks that saturate, needles, faint and giant
Gaussians, ragged image sizes, several views) rendered over a few steps of small random motion through a DepthBoundCache must
equal the unbounded render of the same inputs BIT FOR BIT — image, radii, and every gradient — whether the bound hits, misses
(transparent re-run) or is re-used. usage: soak_depth_bound.py [n_scenes] [seed] [motion scale]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
motion = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0        # scales the per-step motion (large values force misses)
dev = torch.device("cuda:0")
bad = calls = bounded = misses = 0
d_b = d_u = 0
for it in range(n):
    mode = it % 5
    P = rnd.choice([2000, 6000, 20000])
    nv = rnd.randint(1, 3)
    use_rgb = rnd.random() < 0.7
    sc = make_scene("random1k", n_views=nv, P=P, use_rgb=use_rgb, blend=rnd.random() < 0.5, seed=rnd.randint(0, 10**6))
    g = torch.Generator().manual_seed(it)
    if mode == 0:      # a dense opaque slab in front of everything: tiles saturate early, most instances are hidden
        k = P // 2
        sc.xyz[:k, 2] = -0.08 + 0.01 * torch.rand(k, generator=g)
        sc.opacity[:k] = 0.6 + 0.39 * torch.rand(k, 1, generator=g)
        sc.scaling[:k] = 10 ** (-2.2 + 0.3 * torch.rand(k, 3, generator=g))
    elif mode == 1:    # layered shells
        for j in range(4):
            sl = slice(j * (P // 4), (j + 1) * (P // 4))
            sc.xyz[sl, 2] = -0.09 + 0.05 * j + 0.004 * torch.rand(P // 4, generator=g)
        sc.opacity = 0.3 + 0.6 * torch.rand(P, 1, generator=g)
        sc.scaling = 10 ** (-2.3 + 0.4 * torch.rand(P, 3, generator=g))
    elif mode == 2:    # needles + giants over a slab
        sc.scaling = 10 ** (-3.5 + 2.5 * torch.rand(P, 3, generator=g))
        sc.opacity = torch.sigmoid(2 * torch.randn(P, 1, generator=g))
    elif mode == 3:    # opacities hugging the thresholds: pixels that end near T = 1e-4 / alpha = 1/255
        sc.opacity = torch.where(torch.rand(P, 1, generator=g) < 0.5, torch.full((P, 1), 1 / 255 * 1.02), 0.9 * torch.ones(P, 1))
        sc.scaling = 10 ** (-2.4 + 0.5 * torch.rand(P, 3, generator=g))
    else:              # the plain random scene, thicker Gaussians
        sc.scaling = sc.scaling * rnd.choice([1.0, 2.0, 4.0])
    sc.H, sc.W = rnd.randint(24, 200), rnd.randint(24, 200)
    s = sc.to(dev)
    cams = sc.cams().to(dev)
    bl = {k: getattr(s, k) for k in ("xyz_b", "opacity_b", "color_w", "color_b") if getattr(s, k) is not None}
    kw = dict(colors_precomp=s.shs.squeeze(1)) if use_rgb else dict(shs=s.shs, sh_degree=sc.sh_degree)
    cache = R.DepthBoundCache(margin=rnd.choice([5e-4, 2e-3]), slack=rnd.choice([0, 8]), refresh_every=rnd.choice([1, 2, 3]), min_pixels=0)
    xyz = s.xyz.clone()
    sigma = motion * rnd.choice([0.0, 2e-5, 2e-4])
    for step in range(5):
        xyz = xyz + sigma * torch.randn(xyz.shape, generator=g).to(dev)
        dimg = torch.randn(nv, 3, sc.H, sc.W, generator=g).to(dev)
        out = []
        for c in (cache, None):
            img, radii, ctx = R.raster_forward(cams, xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=True, depth_bound=c, **kw, **bl)
            D = R.last_num_rendered()
            gr = R.raster_backward(ctx, dimg, want_means2D=False)
            out.append((img, radii, gr, D))
        (ib, rb, gb, Db), (iu, ru, gu, Du) = out
        calls += 1
        d_b += Db; d_u += Du
        same = torch.equal(ib, iu) and torch.equal(rb, ru) and all(torch.equal(gb[k], gu[k]) for k in gu)
        if not same:
            bad += 1
            worst = max(float((gb[k] - gu[k]).abs().max()) for k in gu)
            print(f"MISMATCH scene {it} mode {mode} step {step} P={P} {sc.H}x{sc.W} nv={nv}: image {float((ib - iu).abs().max()):.3e} grads {worst:.3e}")
    bounded += cache.bounded_calls
    misses += cache.misses
print(f"{n} scenes x 5 steps: {bad} mismatching steps of {calls}; {bounded} bounded calls, {misses} misses (re-run unbounded); "
      f"instances with the bound {d_b} / without {d_u} = {d_b / max(d_u, 1):.2f}")
sys.exit(1 if bad else 0)
