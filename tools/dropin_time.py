"""Where the host time of one drop-in rasteriser call goes (perf_counter around the statements of _RasterizeGaussians.forward)."""
import os, sys, time, weakref
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.camera import Camera, pack_camera
from guassianhand_amd.renderer import GaussianModel
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W)
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
T = {}
def tick(name, t0):
    t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + (t1 - t0); return t1

orig_fwd = R.raster_forward
def timed_raster_forward(*a, **k):
    t0 = time.perf_counter(); r = orig_fwd(*a, **k); tick("raster_forward", t0); return r
orig_bwd = R.raster_backward
def timed_raster_backward(*a, **k):
    t0 = time.perf_counter(); r = orig_bwd(*a, **k); tick("raster_backward", t0); return r
R.raster_forward = timed_raster_forward
R.raster_backward = timed_raster_backward
orig_apply = R._RasterizeGaussians.apply
def timed_apply(*a):
    t0 = time.perf_counter(); r = orig_apply(*a); tick("Function.apply (incl. raster_forward)", t0); return r
R._RasterizeGaussians.apply = timed_apply

def step():
    gs.xyz.grad = None
    t0 = time.perf_counter()
    out = forward_single_view(gs, cam, sc.bg, **kw)
    t0 = tick("forward_single_view", t0)
    loss = out["comp_rgb"].mean() + out["comp_mask"].mean()
    t0 = tick("loss", t0)
    loss.backward()
    tick("backward()", t0)
for _ in range(5): step()
torch.cuda.synchronize(); T.clear()
n = 300
t0 = time.perf_counter()
for _ in range(n): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per view: wall {1e3 * t_all / n:.3f} ms, host enqueue {1e3 * t_enq / n:.3f} ms")
for k, v in T.items():
    print(f"  {k:45s} {1e6 * v / n:8.1f} us per view")
