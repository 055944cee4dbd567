"""Do the scheduling hints of the workspace (launch order by the previous call's measurements, backward work list by the previous
call's costs) still pay when the cameras CHANGE between calls? 8-view steps over a ring of 32 cameras: static (the same 8 every step,
the bench's case), rotating (every view slot moves on by one camera = 11 degrees per step) and random (8 cameras drawn per step).
Eager steps (GPU-bound at 8 views), HIP events around 60 steps. usage: moving_cameras.py  (run once per setting of GH_FWD_HEAVY_ORDER /
GH_BWD_CLASSES: the library reads them once per process)"""
import os, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from guassianhand_amd.scenes import make_scene, ring_cameras
from guassianhand_amd.rasterizer import raster_forward, raster_backward
from guassianhand_amd.camera import pack_cameras_from_w2c
from tests.helpers import scene_kwargs, dimg_like
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8)
s = sc.to(dev); kw, bl = scene_kwargs(s)
H, W = sc.H, sc.W
ring = make_scene("two_hands", n_views=32)
cams_all = ring.cams().to(dev)                     # (32, GH_CAM_FLOATS)
dimg = dimg_like(8, H, W).to(dev)
g = torch.Generator().manual_seed(0)
def run(mode, steps=60):
    idx = torch.arange(8) * 4
    def step(k):
        if mode == "static": ids = idx
        elif mode == "rotating": ids = (idx + k) % 32
        else: ids = torch.randperm(32, generator=g)[:8]
        cams = cams_all[ids.to(dev)]
        img, radii, ctx = raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, sync=False, expect_backward=True, **kw, **bl)
        raster_backward(ctx, dimg)
    for k in range(15): step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(steps): step(15 + k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
print(f"GH_FWD_HEAVY_ORDER={os.environ.get('GH_FWD_HEAVY_ORDER', '1')} GH_BWD_CLASSES={os.environ.get('GH_BWD_CLASSES', '1')}: "
      + ", ".join(f"{m} {run(m):.4f} ms" for m in ("static", "rotating", "random", "static")))
