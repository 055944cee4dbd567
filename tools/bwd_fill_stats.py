"""Distribution of the backward's compacted batch sizes on the bench workload: for every (tile, depth segment, 4x4 block)
the number h of list entries inside the segment's walked part whose block mask reaches the block. Prints a histogram of h
and the lane fill of a few batching policies. Usage: python tools/bwd_fill_stats.py [views] [segment]"""
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
rng = wv["ranges"].long()                              # (T,2)
T = rng.shape[0]
tile_of = wv["sorted_tile"][:D].long()
pos = torch.arange(D, device=dev) - rng[tile_of, 0]
mask = wv["inst_r2"][:D, 1].long() & 0xFFFF
nc = wv["n_contrib"].long()                            # (NV,H,W)
H, W = sc.H, sc.W
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = torch.zeros(nv, gy * 16, gx * 16, dtype=torch.long, device=dev)
pad[:, :H, :W] = nc
blast = pad.reshape(nv, gy, 4, 4, gx, 4, 4).amax(dim=(3, 6))        # (nv, gy, 4(by), gx, 4(bx))
blast = blast.permute(0, 1, 3, 2, 4).reshape(T, 16)                  # block index = by*4+bx
walked = blast.amax(1)
print(f"D {D}  walked {int(walked.sum())}  tiles with work {int((walked > 0).sum())} / {T}")
seg = pos // SEG
tot_pairs = 0
hs = []
for b in range(16):
    hit = ((mask >> b) & 1).bool() & (pos < blast[tile_of, b])
    key = tile_of[hit] * 64 + seg[hit]
    cnt = torch.bincount(key)
    cnt = cnt[cnt > 0]
    hs.append(cnt)
    tot_pairs += int(hit.sum())
h = torch.cat(hs)
print(f"(block, entry) pairs {tot_pairs}; (item, block) units {h.numel()}; mean h {h.float().mean():.1f}")
edges = [0, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (h > lo) & (h <= hi)
    print(f"  h in ({lo},{hi}]: {int(m.sum()):7d} units, {int(h[m].sum()):9d} pairs")
def lanes(policy):
    if policy == "64":
        return ((h + 63) // 64 * 64).sum()
    if policy == "adaptive":                            # 16 / 32 / 64-lane batches for the remainder
        full = h // 64 * 64
        r = h % 64
        rem = torch.where(r == 0, 0, torch.where(r <= 16, 16, torch.where(r <= 32, 32, 64)))
        return (full + rem).sum()
print("lane fill, 64-lane batches:", float(tot_pairs) / float(lanes("64")))
print("lane fill, 16/32/64 adaptive remainder:", float(tot_pairs) / float(lanes("adaptive")))
# quadrant-level units (4 blocks walked one after the other by one wave): batches per wave-item
