"""Per-view render times and per-tile walked-length statistics (is the render kernel tail-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R, _abi
from guassianhand_amd.scenes import make_scene
import ctypes as C
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8).to(dev)
cams = sc.cams()
blend = dict(xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=sc.color_b)
col = dict(colors_precomp=sc.shs.reshape(sc.P, 3))
def run(idx):
    c = cams[idx].contiguous()
    img, radii, ctx = R.raster_forward(c, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True, **col, **blend)
    d = torch.randn_like(img)
    R.raster_backward(ctx, d, want_means2D=False)
    R.enable_stage_timing(True)
    for _ in range(5):
        img, radii, ctx = R.raster_forward(c, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=False, **col, **blend)
        R.raster_backward(ctx, d, want_means2D=False)
    st = R.stage_timing_summary(); R.enable_stage_timing(False)
    wv = R.workspace_views(ctx)
    rng = wv["ranges"].long(); ln = (rng[:, 1] - rng[:, 0])
    L = _lib_layout(ctx)
    walk = ctx.ws[L.tile_walk:L.tile_walk + 4 * ln.numel()].view(torch.int32).long()
    return st, ln, walk
def _lib_layout(ctx):
    from guassianhand_amd import _lib
    lay = _abi.GhLayout(); _lib.lib().gh_workspace_layout(C.byref(ctx.dims), C.byref(lay)); return lay
for name, idx in [("v%d" % v, [v]) for v in range(8)] + [("v0-3", [0, 1, 2, 3]), ("v4-7", [4, 5, 6, 7]), ("all", list(range(8)))]:
    st, ln, walk = run(idx)
    print(f"{name}: fwd {st['render_fwd']:.3f} bwd {st['render_bwd']:.3f} ms | list len max {int(ln.max())} mean {float(ln.float().mean()):.0f} | walked max {int(walk.max())} mean {float(walk.float().mean()):.0f} sum {int(walk.sum())}")
