import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
nv = 8
sc = make_scene("two_hands", n_views=nv, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
modes = [m == "1" for m in (sys.argv[1] if len(sys.argv) > 1 else "01")]
for static in modes:
    f = F.OneShotFit(gs, uv, static_geometry=static)
    with torch.no_grad():
        out = f.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, f.blend_values())
    gt_rgb, gt_mask = (out["comp_rgb"] * 0.9).clone(), out["comp_mask"].mean(-1).clone()
    args = (sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask)
    for i in range(3): f.step(*args, sync=(i == 0))
    ls = [float(f.step(*args, sync=False)) for _ in range(33)]
    print(static, "eager losses", ls[0], ls[-1], "nan at", [i for i, l in enumerate(ls) if l != l][:3])
    R.check_overflow()
    cap = f.captured(*args)
    ls = [float(cap.replay()) for _ in range(33)]
    print(static, "replay losses", ls[0], ls[-1], "nan at", [i for i, l in enumerate(ls) if l != l][:3],
          "counters", [c[0].tolist() for c in cap.counters])
    ob = f.blend_values()["opacity_b"].reshape(-1)
    print("   max op", float((gs.opacity.reshape(-1) + ob).max()), "opacity_b finite", bool(torch.isfinite(ob).all()),
          "color_w finite", bool(torch.isfinite(f.color_w).all()), "color_b finite", bool(torch.isfinite(f.color_b_tex).all()))
    cap.check()
    del f, cap
    torch.cuda.empty_cache()
