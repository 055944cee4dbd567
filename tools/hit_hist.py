"""Histogram of hits per (64-entry batch, 4x4 block) over the walked part of all tile lists (two_hands, 8 views):
how many 4-entry trips could pair up into 8-entry trips."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from guassianhand_amd import rasterizer as R, _abi, _lib
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8).to(dev)
blend = dict(xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=sc.color_b)
img, radii, ctx = R.raster_forward(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True,
                                   colors_precomp=sc.shs.reshape(sc.P, 3), **blend)
wv = R.workspace_views(ctx)
lay = _abi.GhLayout(); _lib.lib().gh_workspace_layout(C.byref(ctx.dims), C.byref(lay))
T = wv["ranges"].shape[0]
walk = ctx.ws[lay.tile_walk:lay.tile_walk + 4 * T].view(torch.int32).long()
rng = wv["ranges"].long()
D = int(wv["counters"][0])
m = (wv["inst_r2"][:D, 1].long() & 0xFFFF)
pos = torch.arange(D, device=dev)
tile_of = torch.repeat_interleave(torch.arange(T, device=dev), rng[:, 1] - rng[:, 0])
local = pos - rng[tile_of, 0]
walked = local < walk[tile_of]
batch_id = tile_of * 4096 + local // 64            # unique per (tile, batch)
bits = torch.stack([(m >> b) & 1 for b in range(16)], 1) * walked[:, None]
uniq, inv = torch.unique(batch_id, return_inverse=True)
hb = torch.zeros(uniq.numel(), 16, dtype=torch.long, device=dev).index_add_(0, inv, bits)
h = hb.flatten(); h = h[h > 0]
trips4 = ((h + 3) // 4).sum()
packed = (h // 8).sum(); rem = h % 8
single = ((rem > 0) & (rem <= 4)).sum(); partial8 = (rem > 4).sum()
print("batch-waves with hits:", h.numel(), "mean hits", float(h.float().mean()))
print("4-wide trips now:", int(trips4))
print("8-wide design: full-8 trips", int(packed), "partial-8 trips (5..7 hits)", int(partial8), "single 4-wide trips", int(single))
print("hits histogram (1..16+):", torch.bincount(h.clamp(max=17))[1:].tolist())
hits = int(h.sum())
print("slot utilisation of the 4-wide trips:", hits / (4.0 * int(trips4)), " hits", hits, " ideal trips (cross-batch packing)", (hits + 3) // 4)
