"""Full-size one-shot fit step (8 views, P = 98,562, 1024x2048 maps, active texels): eager enqueue vs fit.CapturedFitStep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
g = torch.Generator().manual_seed(1)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gt_rgb = torch.rand(8, sc.H, sc.W, 3, generator=g).to(dev)
gt_mask = (torch.rand(8, sc.H, sc.W, generator=g) > 0.5).float().to(dev)
args = (sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask)
f = F.OneShotFit(gs, uv, use_rgb=True)
f.step(*args, sync=True)
for _ in range(5):
    f.step(*args, sync=False)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    f.step(*args, sync=False)
torch.cuda.synchronize()
print(f"fit step, eager enqueue: {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
R.check_overflow()
cap = f.captured(*args)
for _ in range(5):
    cap.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    cap.replay()
torch.cuda.synchronize()
print(f"fit step, CapturedFitStep.replay(): {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
cap.check()
print("loss", float(cap.loss))
