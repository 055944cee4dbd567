import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene, perturbed_target_xyz
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=V); s = sc.to(dev); cams = s.cams()
blend = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
with torch.no_grad():
    gt, _ = R.rasterize_views(cams, perturbed_target_xyz(sc).to(dev), s.opacity, s.scaling, s.rotation, s.shs, H=s.H, W=s.W, use_rgb=True, sync=True, **blend)
params = {k: getattr(s, k).clone().requires_grad_(True) for k in ("xyz", "opacity", "scaling", "rotation", "shs", "xyz_b", "opacity_b", "color_w", "color_b")}
def step(sync):
    for p in params.values(): p.grad = None
    img, _ = R.rasterize_views(cams, params["xyz"], params["opacity"], params["scaling"], params["rotation"], params["shs"], H=s.H, W=s.W, use_rgb=True, sync=sync,
                               xyz_b=params["xyz_b"], opacity_b=params["opacity_b"], color_w=params["color_w"], color_b=params["color_b"])
    loss = (img - gt).abs().mean(); loss.backward(); return loss
step(True); step(False); R.check_overflow()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step(False)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
ref = {k: v.grad.clone() for k, v in params.items()}
# capture
R.set_graph_mode(True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step(False)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step(False)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
ok = all(torch.equal(params[k].grad, ref[k]) for k in params)
print(f"V={V} eager {te*1e3:.3f} ms/step  graph {tg*1e3:.3f} ms/step  grads identical: {ok}  loss {float(loss):.6f}")
