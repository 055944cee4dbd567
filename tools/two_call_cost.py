"""Per-view cost of the reference's own protocol — per view the attribute blend in torch, the RGB pass and the mask pass
through the drop-in GaussianRasterizer, one loss and one backward per step — against the fused batched form. The protocol's harness
is bench.two_call_cost (ONE harness for the bench line and for this tool: rounds 3-5 had two that measured different protocols,
see profiles/r6_two_call_reconcile.txt); this tool adds the step shapes side by side and the fused form."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from guassianhand_amd import rasterizer as R
from guassianhand_amd.renderer import GaussianModel, render_views
from guassianhand_amd.scenes import make_scene

dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8).to(dev)
gs = GaussianModel(sc.xyz.clone().requires_grad_(True), sc.opacity, sc.rotation, sc.scaling, sc.shs)


def fused():
    gs.xyz.grad = None
    out = render_views(gs, sc.w2c, sc.K, sc.H, sc.W, sc.bg, color_w=sc.color_w, xyz_b=sc.xyz_b, color_b=sc.color_b,
                       opacity_b=sc.opacity_b, use_rgb=True, sync=False)
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()


def t(fn, n=30):
    import gc
    for _ in range(4):
        fn()
    gc.collect()          # a pending full collection (40-60 ms for the interpreter's ~1e6 objects) is not part of a 30-call figure
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a8 = bench.two_call_cost(sc, list(range(8)), n_iter=8)
a1 = bench.two_call_cost(sc, [0])
b = t(fused)
R.check_overflow()
print(f"reference protocol through the drop-in, 8 views per step (16 rasteriser calls, one loss, one backward; all leaves): {a8:.3f} ms per view fwd+bwd")
print(f"reference protocol through the drop-in, 1 view per step (2 rasteriser calls, a loss and a backward of its own): {a1:.3f} ms per view fwd+bwd")
print(f"fused batched form (8 views, RGB+alpha in one pass, sync-free): {b:.3f} ms = {b / 8:.3f} ms per view fwd+bwd")
