"""Which _ptr() calls are slow inside the real drop-in flow? (round 5: raster_forward regressed 132 -> 400 us per view)"""
import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctypes as C
from guassianhand_amd import rasterizer as R
from guassianhand_amd.camera import Camera
from guassianhand_amd.renderer import GaussianModel
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W)
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
acc = {}
def timed_ptr(t):
    if t is None:
        return None
    t0 = time.perf_counter()
    a = t.data_ptr()
    t1 = time.perf_counter()
    p = C.c_void_p(a)
    t2 = time.perf_counter()
    k = (tuple(t.shape), str(t.dtype)[6:], t.requires_grad, type(t.grad_fn).__name__ if t.grad_fn is not None else "leaf")
    e = acc.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += t1 - t0; e[2] += t2 - t1
    return p
R._ptr = timed_ptr
def step():
    gs.xyz.grad = None
    out = forward_single_view(gs, cam, sc.bg, **kw)
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
for _ in range(10): step()
torch.cuda.synchronize(); acc.clear()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
print(f"step {1e6 * (time.perf_counter() - t0) / n:.1f} us (host)")
for k, (c, a, b) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(k):90s} calls/step {c / n:5.1f}  data_ptr {1e6 * a / c:8.2f} us  c_void_p {1e6 * b / c:6.2f} us")
