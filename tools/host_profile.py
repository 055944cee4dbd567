"""Where the host time of one forward+backward step goes (cProfile of the enqueue path; GPU box).
usage: host_profile.py [views_per_step]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.loss import l1_mean_loss
from guassianhand_amd.scenes import make_scene

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
s = make_scene("two_hands", n_views=V).to(dev)
cams = s.cams().contiguous()
blend = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
with torch.no_grad():
    gt, _ = R.rasterize_views(cams, s.xyz, s.opacity, s.scaling, s.rotation, s.shs, H=s.H, W=s.W, use_rgb=True, sync=True, **blend)
params = {"xyz": s.xyz, "opacity": s.opacity, "scaling": s.scaling, "rotation": s.rotation, "shs": s.shs, **blend}
params = {k: v.clone().requires_grad_(True) for k, v in params.items()}


def step(sync=False):
    for p in params.values():
        p.grad = None
    img, _ = R.rasterize_views(cams, params["xyz"], params["opacity"], params["scaling"], params["rotation"], params["shs"],
                               H=s.H, W=s.W, use_rgb=True, sync=sync, xyz_b=params["xyz_b"], opacity_b=params["opacity_b"],
                               color_w=params["color_w"], color_b=params["color_b"])
    loss = l1_mean_loss(img, gt)
    loss.backward()


step(True)
for _ in range(5):
    step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"V={V}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
