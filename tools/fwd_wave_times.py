"""Timing experiment (library built with -DGH_EXP_WTIME: tools/abl_build.sh wtime "-DGH_EXP_WTIME"; GH_RASTER_LIB=tools/abl/wtime.so):
every forward wave leaves (start cycle, duration, batches << 16 | trips, list length | tile << 16) behind; which waves make the
one-view forward as long as it is?  usage: python tools/fwd_wave_times.py [views]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from guassianhand_amd import _abi, _lib
from guassianhand_amd.rasterizer import raster_forward
from guassianhand_amd.scenes import make_scene

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=V).to(dev)
cams = sc.cams().contiguous()
kw = dict(H=sc.H, W=sc.W, colors_precomp=sc.shs.squeeze(1), xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=sc.color_b)
for _ in range(3):
    img, _, ctx = raster_forward(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, **kw)
torch.cuda.synchronize()
lay = _abi.GhLayout()
_lib.lib().gh_workspace_layout(C.byref(ctx.dims), C.byref(lay))
tiles = V * ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
nw = tiles * 16
grid = tiles * 4
rec = ctx.ws[lay.final_C + grid * 4 * 16: lay.final_C + grid * 4 * 16 + nw * 16].view(torch.int32).reshape(nw, 4).cpu().long() & 0xFFFFFFFF
start, dur, bt, tt = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
trips, batches, total, tile = bt & 0xFFFF, bt >> 16, tt & 0xFFFF, tt >> 16
t0 = int(start.min())
end = (start - t0) + dur
print(f"{V} view(s): {nw} waves; kernel span {int(end.max())} cycles (s_memtime units); sum of wave durations {int(dur.sum())}")
order = torch.argsort(end, descending=True)[:12]
print(" last waves to finish:  start    dur   end  | batches trips list | cycles/trip cycles/batch | tile")
for i in order.tolist():
    print(f"   {int(start[i]-t0):8d} {int(dur[i]):6d} {int(end[i]):6d} | {int(batches[i]):5d} {int(trips[i]):5d} {int(total[i]):5d} | "
          f"{int(dur[i]) / max(1, int(trips[i])):8.1f} {int(dur[i]) / max(1, int(batches[i])):8.1f} | {int(tile[i])}")
lo = torch.argsort(dur, descending=True)[:8]
print(" longest waves:")
for i in lo.tolist():
    print(f"   {int(start[i]-t0):8d} {int(dur[i]):6d} {int(end[i]):6d} | {int(batches[i]):5d} {int(trips[i]):5d} {int(total[i]):5d} | "
          f"{int(dur[i]) / max(1, int(trips[i])):8.1f} {int(dur[i]) / max(1, int(batches[i])):8.1f} | {int(tile[i])}")
print(f" all waves: trips {int(trips.sum())}, batches {int(batches.sum())}; mean cycles per trip over waves with >= 50 trips: "
      f"{float((dur[trips >= 50].double() / trips[trips >= 50].double()).mean()):.1f}")
import collections
h = collections.Counter((start - t0).div(2000, rounding_mode='floor').tolist())
print(" wave starts per 2000-cycle bin:", dict(sorted(h.items())))
