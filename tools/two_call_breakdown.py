"""Host time of the library calls inside the reference protocol (two rasteriser calls per view, one backward over all views):
perf_counter around every C-ABI call, by entry point. (With docs/history/r4_replay_experiment.patch applied it also reports what
that experiment replayed: GH_RASTER_REPLAY=0/1.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import _lib, rasterizer as R
from guassianhand_amd.camera import Camera
from guassianhand_amd.renderer import GaussianModel
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = make_scene("two_hands", n_views=NV).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cams = [Camera.from_w2c(sc.w2c[v], sc.K[v], sc.H, sc.W) for v in range(NV)]
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
L = _lib.lib()
T, N = {}, {}
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); dt = time.perf_counter() - t0
        T[self.name] = T.get(self.name, 0.0) + dt; N[self.name] = N.get(self.name, 0) + 1
        return r
for name in ("gh_forward", "gh_backward", "gh_forward_shared", "gh_backward_shared", "gh_forward_refresh", "gh_backward_refresh"):
    setattr(L, name, Timed(name, getattr(L, name)))
def step():
    gs.xyz.grad = None
    loss = 0
    for v in range(NV):
        out = forward_single_view(gs, cams[v], sc.bg, **kw)
        loss = loss + out["comp_rgb"].mean() + out["comp_mask"].mean()
    loss.backward()
for _ in range(4): step()
torch.cuda.synchronize(); T.clear(); N.clear()
n = 30
t0 = time.perf_counter()
for _ in range(n): step()
enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
rs = f"; GH_RASTER_REPLAY={os.environ.get('GH_RASTER_REPLAY', '1')}, replayed / captured {R.replay_stats()}" if hasattr(R, "replay_stats") else ""
print(f"{NV} views: wall {1e3 * wall / n / NV:.3f} ms per view, host enqueue {1e3 * enq / n / NV:.3f} ms per view{rs}")
for k in T:
    print(f"  {k:22s} {N[k] / n / NV:4.1f} calls per view, {1e6 * T[k] / N[k]:7.1f} us per call")
