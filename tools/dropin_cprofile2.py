"""cProfile of the drop-in protocol: every function with >= 0.3 ms of own time over 200 steps, and the callers of the heaviest."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for _ in range(10): step()
torch.cuda.synchronize()
import gc
print("gc thresholds", gc.get_threshold(), "counts", gc.get_count())
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); ps = pstats.Stats(pr, stream=st).sort_stats("tottime"); ps.print_stats(60); print(st.getvalue()[:9000])
st = io.StringIO(); ps = pstats.Stats(pr, stream=st); ps.print_callees("_inputs_struct"); ps.print_callees("_forward_shared"); print(st.getvalue()[:6000])
