"""Per (tile, depth segment) work item of the render backward: how many list entries reach each 8x8 quadrant (the rows of
the quadrant's accumulator), and how many (entry, quadrant) sub-records against (entry) records the item produces.
Usage: python tools/bwd_quad_rows.py [views] [segment]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
from tests.helpers import scene_kwargs

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SEG = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=nv)
s = sc.to(dev)
kw, bl = scene_kwargs(s)
img, _, ctx = R.raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
wv = R.workspace_views(ctx)
D = int(wv["counters"][0])
rng = wv["ranges"].long()
T = rng.shape[0]
tile_of = wv["sorted_tile"][:D].long()
pos = torch.arange(D, device=dev) - rng[tile_of, 0]
mask = wv["inst_r2"][:D, 1].long() & 0xFFFF
nc = wv["n_contrib"].long()
H, W = sc.H, sc.W
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = torch.zeros(nv, gy * 16, gx * 16, dtype=torch.long, device=dev)
pad[:, :H, :W] = nc
qlast = pad.reshape(nv, gy, 2, 8, gx, 2, 8).amax(dim=(3, 6)).permute(0, 1, 3, 2, 4).reshape(T, 4)    # quadrant = qy*2+qx
walked = qlast.amax(1)
seg = pos // SEG
nseg = int(seg.max()) + 1
rows = torch.zeros(T * nseg, 4, dtype=torch.long, device=dev)
anyq = torch.zeros(D, dtype=torch.bool, device=dev)
for q in range(4):
    qbits = 0x33 << (8 * (q >> 1) + 2 * (q & 1))
    hit = ((mask & qbits) != 0) & (pos < qlast[tile_of, q])
    anyq |= hit
    rows[:, q] = torch.bincount((tile_of * nseg + seg)[hit], minlength=T * nseg)
live = rows.sum(1) > 0
rows = rows[live]
mx = rows.amax(1)
print(f"D {D} walked {int(walked.sum())} items {int(live.sum())} (segment {SEG})")
print(f"(entry, quadrant) sub-records {int(rows.sum())}  entries with a record {int(anyq.sum())}  ratio {float(rows.sum()) / float(anyq.sum()):.2f}")
print(f"rows per (item, quadrant): mean {rows.float().mean():.1f}; max over the 4 quadrants: mean {mx.float().mean():.1f}")
for lim in (64, 96, 128, 160, 192, 224, 256):
    print(f"  items with max rows <= {lim}: {float((mx <= lim).float().mean()) * 100:.1f} %")
nzq = (rows > 0).sum(1)
for k in range(1, 5):
    print(f"  items with {k} non-empty quadrants: {float((nzq == k).float().mean()) * 100:.1f} %")
imb = rows.amax(1).float() / rows.float().mean(1).clamp(min=1)
print(f"quadrant imbalance max/mean: mean {imb.mean():.2f}")
print(f"sum over items of max rows x 4 / sum rows = {float(mx.sum() * 4) / float(rows.sum()):.2f}  (work held by a 4-wave group vs its useful rows)")
