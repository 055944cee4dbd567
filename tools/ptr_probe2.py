"""Host time of the drop-in protocol per step under variations (round 5: raster_forward regressed): which one removes the slowness?
usage: python tools/ptr_probe2.py <variant>   variants: plain | nogc | sameptr | freeze"""
import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctypes as C
from guassianhand_amd import rasterizer as R
from guassianhand_amd.camera import Camera
from guassianhand_amd.renderer import GaussianModel
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W)
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
if variant == "sameptr":
    def _ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())
    R._ptr = _ptr
def step():
    gs.xyz.grad = None
    out = forward_single_view(gs, cam, sc.bg, **kw)
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
for _ in range(10): step()
torch.cuda.synchronize()
if variant == "nogc":
    gc.disable()
if variant == "freeze":
    gc.collect(); gc.freeze()
n = 300
c0 = gc.get_stats()
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
c1 = gc.get_stats()
print(f"{variant:8s}: host {1e6 * (t1 - t0) / n:7.1f} us per step, wall {1e6 * (time.perf_counter() - t0) / n:7.1f}; gc collections per generation during the loop: "
      f"{[b['collections'] - a['collections'] for a, b in zip(c0, c1)]}")
