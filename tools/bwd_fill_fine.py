"""What a finer block mask would buy the render backward: for the bench workload, the number of (entry, block) pairs and
lane-iterations with the present 4x4-pixel blocks (16 iterations per pair) against 4x2-pixel blocks (8 iterations per pair),
using the same ellipse / rectangle test (two facing edges) evaluated in torch, restricted to the walked part of every list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from guassianhand_amd import rasterizer as R, _abi, _lib
from guassianhand_amd.scenes import make_scene
from tests.helpers import scene_kwargs

nv = 8
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=nv)
s = sc.to(dev)
kw, bl = scene_kwargs(s)
img, _, ctx = R.raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, split_streams=False, **kw, **bl)
wv = R.workspace_views(ctx)
L = _lib.lib(); lay = _abi.GhLayout(); L.gh_workspace_layout(C.byref(ctx.dims), C.byref(lay))
D = int(wv["counters"][0])
cap = int(ctx.dims.max_instances)
r0 = ctx.ws[lay.inst_r0:lay.inst_r0 + cap * 16].view(torch.float32).reshape(cap, 4)[:D]
r1 = ctx.ws[lay.inst_r1:lay.inst_r1 + cap * 16].view(torch.float32).reshape(cap, 4)[:D]
px, py, A, B = r0[:, 0], r0[:, 1], r0[:, 2], r0[:, 3]
Cc, op = r1[:, 0], r1[:, 1]
rng = wv["ranges"].long(); T = rng.shape[0]
tile_of = wv["sorted_tile"][:D].long()
pos = torch.arange(D, device=dev) - rng[tile_of, 0]
H, W = sc.H, sc.W
gx, gy = (W + 15) // 16, (H + 15) // 16
tl = tile_of % (gx * gy)
tx0 = (tl % gx).float() * 16; ty0 = (tl // gx).float() * 16
nc = wv["n_contrib"].long()
pad = torch.zeros(nv, gy * 16, gx * 16, dtype=torch.long, device=dev); pad[:, :H, :W] = nc
thr = 2.0 * (torch.log(255.0 * op) * 1.0001 + 1e-3)

def hit(bw, bh):
    """(D, 16/bh, 16/bw) bool: alpha >= 1/255 ellipse reaches a pixel centre of the bw x bh block"""
    nx, ny = 16 // bw, 16 // bh
    lx = (tx0[:, None] + torch.arange(nx, device=dev)[None] * bw - px[:, None])[:, None, :].expand(D, ny, nx)
    ly = (ty0[:, None] + torch.arange(ny, device=dev)[None] * bh - py[:, None])[:, :, None].expand(D, ny, nx)
    ux, uy = lx + (bw - 1), ly + (bh - 1)
    xn = torch.minimum(torch.clamp(lx, min=0), ux); yn = torch.minimum(torch.clamp(ly, min=0), uy)
    a, b, c = A[:, None, None], B[:, None, None], Cc[:, None, None]
    dyb = torch.minimum(torch.maximum(-b * xn / c, ly), uy); dxb = torch.minimum(torch.maximum(-b * yn / a, lx), ux)
    q0 = a * xn * xn + 2 * b * xn * dyb + c * dyb * dyb
    q1 = a * dxb * dxb + 2 * b * dxb * yn + c * yn * yn
    return (torch.minimum(q0, q1) * 0.9999 <= thr[:, None, None]) & (op[:, None, None] >= 1 / 255)

def walked(bw, bh):
    nx, ny = 16 // bw, 16 // bh
    last = pad.reshape(nv, gy, ny, bh, gx, nx, bw).amax(dim=(3, 6))          # (nv, gy, ny, gx, nx)
    last = last.permute(0, 1, 3, 2, 4).reshape(T, ny, nx)
    return pos[:, None, None] < last[tile_of]

for bw, bh in ((4, 4), (4, 2), (2, 2)):
    h = hit(bw, bh) & walked(bw, bh)
    pairs = int(h.sum())
    print(f"{bw}x{bh} blocks: {pairs} (entry, block) pairs in the walked lists -> {pairs * bw * bh / 1e6:.1f} M lane-iterations")
# contributing (entry, pixel) pairs for reference: alpha >= 1/255 at the pixel centre, inside the walked prefix
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
tot = 0
for lo in range(0, D, 200000):
    sl = slice(lo, min(D, lo + 200000))
    dx = px[sl, None, None] - (tx0[sl, None, None] + xx[None]); dy = py[sl, None, None] - (ty0[sl, None, None] + yy[None])
    q = A[sl, None, None] * dx * dx + 2 * B[sl, None, None] * dx * dy + Cc[sl, None, None] * dy * dy
    al = torch.clamp(op[sl, None, None] * torch.exp(-0.5 * q), max=0.99)
    ncp = pad.reshape(nv, gy, 16, gx, 16).permute(0, 1, 3, 2, 4).reshape(T, 16, 16)[tile_of[sl]]
    tot += int(((al >= 1 / 255) & (q >= 0) & (pos[sl, None, None] < ncp)).sum())
print(f"contributing (entry, pixel) pairs: {tot / 1e6:.1f} M")
