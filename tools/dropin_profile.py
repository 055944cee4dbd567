"""Host-side profile of the reference's own per-view protocol through the drop-in (forward_single_view: RGB call + mask call)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.camera import Camera
from guassianhand_amd.renderer import GaussianModel
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W)
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
def step():
    gs.xyz.grad = None
    out = forward_single_view(gs, cam, sc.bg, **kw)
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
for _ in range(5): step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"per view fwd+bwd (2 rasteriser calls): {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
