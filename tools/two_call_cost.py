"""Per-view cost of the reference's own protocol: forward_single_view = RGB pass + mask pass through the
drop-in GaussianRasterizer (one host read-back each, like the reference wrapper), forward + backward, against the fused
batched form."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.camera import Camera
from guassianhand_amd.renderer import GaussianModel, render_views
from tests.helpers import forward_single_view
from guassianhand_amd.scenes import make_scene
from guassianhand_amd import rasterizer as R
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)
cams = [Camera.from_w2c(sc.w2c[v], sc.K[v], sc.H, sc.W) for v in range(8)]
kw = dict(color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b, opacity_b=sc.opacity_b.view(-1, 1), use_rgb=True, sh_degree=3)
def ref_protocol():
    gs.xyz.grad = None
    loss = 0
    for v in range(8):
        out = forward_single_view(gs, cams[v], sc.bg, **kw)
        loss = loss + out["comp_rgb"].mean() + out["comp_mask"].mean()
    loss.backward()
def fused():
    gs.xyz.grad = None
    out = render_views(gs, sc.w2c, sc.K, sc.H, sc.W, sc.bg, color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b,
                       opacity_b=sc.opacity_b, use_rgb=True, sync=False)
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
def t(fn, n=30):
    import gc
    for _ in range(4): fn()
    gc.collect()          # a pending full collection (40-60 ms for the interpreter's ~1e6 objects) is not part of a 30-call figure
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    global enq
    enq = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = t(ref_protocol); a_enq = enq; b = t(fused)
R.check_overflow()
print(f"reference protocol (8 views x 2 rasteriser calls through the drop-in): {a:.3f} ms = {a / 8:.3f} ms per view fwd+bwd (host enqueue {a_enq / 8:.3f})")
print(f"fused batched form (8 views, RGB+alpha in one pass, sync-free): {b:.3f} ms = {b / 8:.3f} ms per view fwd+bwd")
