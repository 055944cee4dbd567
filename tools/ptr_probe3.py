"""Time every statement group of _forward_shared / _forward_full in the real drop-in flow (round 5 host regression hunt)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctypes as C
from guassianhand_amd import rasterizer as R, _abi
from guassianhand_amd.camera import Camera
from guassianhand_amd.renderer import GaussianModel
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W)
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
T = {}
names = ("cams", "means3D", "opacities", "scales", "rotations", "shs", "colors_precomp", "xyz_b", "opacity_b", "color_w", "color_b")
def inputs_struct(c, with_shs=True, bound=None):
    t = c.t
    vals = []
    for k in names:
        t0 = time.perf_counter()
        v = R._ptr(t[k]) if (k != "shs" or with_shs) else None
        T[("ptr", k, with_shs)] = T.get(("ptr", k, with_shs), 0.0) + time.perf_counter() - t0
        vals.append(v)
    t0 = time.perf_counter()
    r = _abi.GhInputs(*vals, R._ptr(bound), R._ptr(t["cov3D"]))
    T[("struct", "", with_shs)] = T.get(("struct", "", with_shs), 0.0) + time.perf_counter() - t0
    return r
R._inputs_struct = inputs_struct
def step():
    gs.xyz.grad = None
    out = forward_single_view(gs, cam, sc.bg, **kw)
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
for _ in range(10): step()
torch.cuda.synchronize(); T.clear()
n = 300
t0 = time.perf_counter()
for _ in range(n): step()
print(f"host {1e6 * (time.perf_counter() - t0) / n:.1f} us per step")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(k, f"{1e6 * v / n:8.2f} us per step")
