"""Soak test of the depth-segmented backward: dense, faint stacks (thousands of blended entries per pixel, many segments
per tile) and mixed scenes; every gradient against the oracle's (double accumulation). usage: soak_grad.py [n] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
from oracle.oracle_c import OracleRender
from tests.helpers import scene_kwargs, dimg_like, rel_l2, max_rel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
dev = torch.device("cuda:0")
worst = {}
bad = 0
for it in range(n):
    P = rnd.choice([4000, 12000, 30000])
    use_rgb = rnd.random() < 0.7
    sc = make_scene("random1k", n_views=rnd.randint(1, 2), P=P, use_rgb=use_rgb, blend=rnd.random() < 0.5, seed=rnd.randint(0, 10**6))
    g = torch.Generator().manual_seed(it)
    sc.H, sc.W = rnd.choice([(32, 48), (64, 64), (96, 40)])
    sc.opacity = (0.004 + 0.03 * torch.rand(P, 1, generator=g)) if it % 2 == 0 else torch.sigmoid(2 * torch.randn(P, 1, generator=g))
    sc.scaling = 10 ** (-2.6 + 0.8 * torch.rand(P, 3, generator=g))            # wide footprints: long lists everywhere
    kw, bl = scene_kwargs(sc)
    o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=True, **kw, **bl)
    s = sc.to(dev)
    kwg, blg = scene_kwargs(s)
    img, radii, ctx = R.raster_forward(sc.cams().to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=True, **kwg, **blg)
    ok = torch.equal(img.cpu(), o.image)
    d = dimg_like(sc.w2c.shape[0], sc.H, sc.W)
    gg = R.raster_backward(ctx, d.to(dev))
    og = o.backward(d)
    walk = int(o.debug["n_contrib"].max())
    for k in og:
        e = rel_l2(gg[k].cpu(), og[k]); m = max_rel(gg[k].cpu(), og[k])
        worst[k] = max(worst.get(k, (0, 0)), (e, m))
        if e > 1e-4: ok = False
    o.close()
    if not ok:
        bad += 1
        print(f"MISMATCH scene {it}: P={P} {sc.H}x{sc.W} max walked {walk}")
    elif it % 10 == 0:
        print(f"scene {it}: P={P} {sc.H}x{sc.W} instances {R.last_num_rendered()} max walked {walk} ok")
print(f"{n} scenes, {bad} bad; worst (rel-L2, max-rel) per gradient:")
for k, v in worst.items(): print(f"  {k}: {v[0]:.2e} {v[1]:.2e}")
sys.exit(1 if bad else 0)
