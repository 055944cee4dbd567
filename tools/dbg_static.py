import sys, os
sys.path.insert(0, os.getcwd())
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = make_scene("two_hands", n_views=nv, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
f = F.OneShotFit(gs, uv, static_geometry=True)
with torch.no_grad():
    out = f.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, f.blend_values())
gt_rgb, gt_mask = (out["comp_rgb"] * 0.9).clone(), out["comp_mask"].mean(-1).clone()
args = (sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask)
for i in range(3):
    print("eager", i, float(f.step(*args, sync=(i == 0))))
cap = f.captured(*args)
print("captured; cache", f._geom_cache.builds, f._geom_cache.hits, "n graph counters", len(cap.counters))
for i in range(12):
    l = float(cap.replay())
    cs = [c[0].tolist() for c in cap.counters]
    ob = f.opacity_b_tex
    print("replay", i, "loss", l, "counters", cs, "opacity_b finite", bool(torch.isfinite(ob).all()), "color_w", f.color_w[:3].tolist(),
          "steps", f._adam["color_w"].step_state.tolist())
    if l != l:
        break
