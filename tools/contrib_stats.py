"""Of the (4x4 block, list entry) pairs the backward evaluates (block-mask bit set, position below the block's last blended
entry), how many blend at no pixel of the block? (two_hands, one view; torch on the GPU, exp from torch: statistics only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from guassianhand_amd import rasterizer as R, _abi, _lib
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = make_scene("two_hands", n_views=8).to(dev)
blend = dict(xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=sc.color_b)
cams = sc.cams()[view:view + 1].contiguous()
img, radii, ctx = R.raster_forward(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True,
                                   colors_precomp=sc.shs.reshape(sc.P, 3), **blend)
wv = R.workspace_views(ctx)
lay = _abi.GhLayout(); _lib.lib().gh_workspace_layout(C.byref(ctx.dims), C.byref(lay))
D = int(wv["counters"][0]); cap = int(ctx.dims.max_instances)
r0 = ctx.ws[lay.inst_r0:lay.inst_r0 + cap * 16].view(torch.float32).reshape(cap, 4)[:D]
r1 = ctx.ws[lay.inst_r1:lay.inst_r1 + cap * 16].view(torch.float32).reshape(cap, 4)[:D]
mask = (wv["inst_r2"][:D, 1].long() & 0xFFFF)
rng = wv["ranges"].long(); T = rng.shape[0]
gx = (sc.W + 15) // 16
nc = wv["n_contrib"][0].long()                                  # (H, W)
tile_of = wv["sorted_tile"][:D].long()
local = torch.arange(D, device=dev) - rng[tile_of, 0]
tx, ty = tile_of % gx, tile_of // gx
tot = contrib = 0
for b in range(16):
    bx, by = b & 3, b >> 2
    sel = ((mask >> b) & 1) == 1
    x0 = (tx * 16 + bx * 4)[sel]; y0 = (ty * 16 + by * 4)[sel]
    px, py, A, Bc = r0[sel, 0], r0[sel, 1], r0[sel, 2], r0[sel, 3]
    Cc, op = r1[sel, 0], r1[sel, 1]
    loc = local[sel]
    anyc = torch.zeros(x0.numel(), dtype=torch.bool, device=dev)
    last_blk = torch.zeros(x0.numel(), dtype=torch.long, device=dev)
    for j in range(16):
        x = x0 + (j & 3); y = y0 + (j >> 2)
        ins = (x < sc.W) & (y < sc.H)
        last = torch.where(ins, nc[y.clamp(max=sc.H - 1), x.clamp(max=sc.W - 1)], torch.zeros_like(x))
        last_blk = torch.maximum(last_blk, last)
        dx = px - x.float(); dy = py - y.float()
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - Bc * dx * dy
        alpha = torch.minimum(torch.full_like(power, 0.99), op * torch.exp(power.clamp(max=0)))
        anyc |= ins & (loc < last) & (power <= 0) & (alpha >= 1 / 255)
    ev = loc < last_blk                                        # what the backward's wave evaluates
    tot += int(ev.sum()); contrib += int((anyc & ev).sum())
print(f"view {view}: (block, entry) pairs evaluated by the backward {tot}, of which blend somewhere {contrib} = {contrib / tot:.3f}")
