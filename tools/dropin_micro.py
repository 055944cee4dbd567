"""Host cost of the pieces of one drop-in call: the bare C entry points (kernel launches inside the library) against the Python
around them (tensor preparation, ctypes structs, allocations, autograd Function plumbing)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import torch
from guassianhand_amd import rasterizer as R, _abi, _lib
from guassianhand_amd.camera import Camera, pack_camera
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1).to(dev)
cams = sc.cams()
L = _lib.lib()
col = sc.shs.reshape(-1, 3).contiguous()


def bench(fn, n=300, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return t * 1e6

img, radii, ctx = R.raster_forward(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, colors_precomp=col, sync=True)
stream = R._raw_stream(dev)
out = _abi.GhOutputs(R._ptr(img), R._ptr(radii), None)
args = (C.byref(ctx.dims), C.byref(ctx.inp), C.byref(out), C.c_void_p(ctx.ws.data_ptr()), ctx.ws.numel(), C.c_void_p(stream))
print(f"bare gh_forward (C call, ~25 launches):            {bench(lambda: L.gh_forward(*args)):7.1f} us")
g = torch.zeros(1, 3, sc.H, sc.W, device=dev)
grads = R.raster_backward(ctx, g)
gr_keep = []
def bare_bwd():
    return R.raster_backward(ctx, g)
print(f"raster_backward (python + C):                      {bench(bare_bwd):7.1f} us")
print(f"raster_forward sync=False (python + C):            {bench(lambda: R.raster_forward(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, colors_precomp=col, sync=False)):7.1f} us")
R.check_overflow()
ones = torch.ones_like(sc.xyz)
print(f"raster_forward geometry_of (shared, python + C):   {bench(lambda: R.raster_forward(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, colors_precomp=ones, sync=False, geometry_of=ctx)):7.1f} us")
cam = Camera.from_w2c(sc.w2c[0], sc.K[0], sc.H, sc.W)
import math
tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
print(f"pack_camera (cached prefix):                       {bench(lambda: pack_camera(cam.world_view_transform, cam.full_proj_transform, cam.camera_center, tfx, tfy, sc.bg)):7.1f} us")
print(f"torch.empty image+radii:                           {bench(lambda: (torch.empty(1, 3, sc.H, sc.W, device=dev), torch.empty(1, sc.P, dtype=torch.int32, device=dev))):7.1f} us")
print(f"GhDims + gh_workspace_bytes:                       {bench(lambda: L.gh_workspace_bytes(C.byref(_abi.GhDims(sc.P, 1, sc.H, sc.W, 0, 0, 1.0, 0, 1000000)))):7.1f} us")
t = [sc.xyz, sc.opacity, sc.scaling, sc.rotation, col, cams]
print(f"11 x _prep + reshape:                              {bench(lambda: [R._prep(x, dev) for x in t] + [R._prep(None, dev)] * 5 + [t[1].reshape(-1), t[5].reshape(-1, 40)]):7.1f} us")
print(f"GhInputs struct of 11 pointers:                    {bench(lambda: _abi.GhInputs(*[R._ptr(x) for x in t] + [None] * 5)):7.1f} us")
host = torch.empty(4, dtype=torch.int32, pin_memory=True); ev = torch.cuda.Event()
cnt = ctx.ws[:16].view(torch.int32)
print(f"pending read-back (copy_ non_blocking + event):    {bench(lambda: (host.copy_(cnt, non_blocking=True), ev.record())):7.1f} us")
