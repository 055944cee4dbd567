"""Soak test of the exact culling margins: many random scenes (needles, faint, giant, border-aligned Gaussians) must
render bit-identically to the oracle, which has no culling. usage: soak_parity.py [n_scenes] [seed]"""
import sys, os, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
from oracle.oracle_c import OracleRender
from tests.helpers import scene_kwargs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
bad = 0
tot_d = tot_r = 0
for it in range(n):
    P = rnd.choice([300, 1000, 3000])
    sc = make_scene("random1k", n_views=rnd.randint(1, 3), P=P, use_rgb=True, blend=False, seed=rnd.randint(0, 10**6))
    g = torch.Generator().manual_seed(it)
    mode = it % 5
    if mode == 0:      # needles at all aspect ratios
        a = 10 ** (-1.5 - 3 * torch.rand(P, generator=g))
        b = a * 10 ** (-3 * torch.rand(P, generator=g))
        sc.scaling = torch.stack([a, b, b], 1)
    elif mode == 1:    # opacities hugging the 1/255 threshold and tiny
        sc.opacity = (1 / 255) * (1 + 0.1 * torch.randn(P, 1, generator=g)).clamp(min=0.5)
    elif mode == 2:    # giants
        sc.scaling = 10 ** (-2.0 + 1.5 * torch.rand(P, 3, generator=g))
        sc.opacity = 0.02 + 0.2 * torch.rand(P, 1, generator=g)
    elif mode == 3:    # centres snapped to pixel / block borders
        q = rnd.choice([1, 4, 16])
        sc.xyz[:, :2] = torch.round(sc.xyz[:, :2] * 325 / q) * q / 325 + rnd.choice([0.0, 0.5 / 325])
        sc.xyz[:, 2] = 0.0
    else:              # everything mixed, wide opacity range
        sc.scaling = 10 ** (-4.5 + 3.5 * torch.rand(P, 3, generator=g))
        sc.opacity = torch.sigmoid(3 * torch.randn(P, 1, generator=g))
    sc.H, sc.W = rnd.randint(16, 140), rnd.randint(16, 140)
    kw, bl = scene_kwargs(sc)
    o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, **kw, **bl)
    s = sc.to(dev)
    kwg, blg = scene_kwargs(s)
    img, radii, ctx = R.raster_forward(sc.cams().to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=True, **kwg, **blg)
    same = torch.equal(img.cpu(), o.image) and torch.equal(radii.cpu(), o.radii)
    tot_d += R.last_num_rendered()
    o.close()
    if not same:
        bad += 1
        print(f"MISMATCH scene {it} mode {mode} P={P} {sc.H}x{sc.W}: max diff {(img.cpu() - o.image).abs().max().item():.3e}")
print(f"{n} scenes, {bad} mismatches, {tot_d} instances rendered")
sys.exit(1 if bad else 0)
