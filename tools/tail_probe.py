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

# heaviest tile of view 3: hits per 4x4 block over the walked prefix of its list
st, ln, walk = run([3])
c = cams[[3]].contiguous()
img, radii, ctx = R.raster_forward(c, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True, **col, **blend)
wv = R.workspace_views(ctx); L = _lib_layout(ctx)
walk = ctx.ws[L.tile_walk:L.tile_walk + 4 * ln.numel()].view(torch.int32).long()
t = int(walk.argmax()); rng = wv["ranges"].long()[t]; w = int(walk[t])
masks = wv["inst_r2"][int(rng[0]):int(rng[0]) + w, 1].long() & 0xFFFF
hits = torch.stack([(masks >> b) & 1 for b in range(16)], 1)          # (walked, 16)
per_block = hits.sum(0)
nb = (w + 63) // 64
pad = torch.zeros(nb * 64 - w, 16, dtype=hits.dtype, device=hits.device)
hb = torch.cat([hits, pad]).view(nb, 64, 16).sum(1)                    # hits per batch per block
trips = (hb + 3) // 4
print(f"heaviest tile {t}: list {int(rng[1]-rng[0])}, walked {w}, batches {nb}")
print("hits per block:", per_block.tolist())
print("trips per block (sum over batches):", trips.sum(0).tolist())
q = trips.view(nb, 2, 2, 2, 2)     # by(2) hi, by lo, bx hi, bx lo -> quadrant = (by hi, bx hi)
quad_max = trips.view(nb, 4, 4)    # rows by, cols bx
# quadrant q=(qy,qx): blocks by in {2qy,2qy+1}, bx in {2qx,2qx+1}; per batch the barrier waits for the slowest of its 4 waves
tot = {}
for qy in range(2):
    for qx in range(2):
        blk = quad_max[:, 2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].reshape(nb, 4)
        tot[(qy, qx)] = (int(blk.max(1).values.sum()), int(blk.float().mean(1).sum()))
print("per quadrant: (sum over batches of max-wave trips, of mean-wave trips):", tot)
