"""Run 20 active-texel fit steps (full size) — target of `rocprofv3 --kernel-trace --stats`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
f = F.OneShotFit(gs, uv)
with torch.no_grad():
    out = f.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, f.blend_values())
gt_rgb, gt_mask = (out["comp_rgb"] * 0.9).clone(), out["comp_mask"].mean(-1).clone()
for i in range(3): f.step(sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask, sync=(i == 0))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): f.step(sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask, sync=False)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"fit step {(t2 - t0) / 20 * 1e3:.3f} ms, host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms")
R.check_overflow()
