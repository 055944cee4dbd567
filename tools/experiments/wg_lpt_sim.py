"""How much of a kernel is the ORDER of its workgroups? Per-workgroup cycle counters (library built with -DGH_EXP_WG_TIME:
bash tools/abl_build.sh wgtime -DGH_EXP_WG_TIME; GH_RASTER_LIB=tools/abl/wgtime.so) of the instrumented kernels over one step of the
bench's default scene, then greedy list scheduling of the measured durations over the kernel's workgroup slots: in dispatch order
(= workgroup index), longest first, and the lower bounds. usage: wg_lpt_sim.py [views] [config]"""
import ctypes as C, heapq, sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from guassianhand_amd import _lib
from guassianhand_amd.scenes import make_scene
from guassianhand_amd.rasterizer import raster_forward, raster_backward
from tests.helpers import scene_kwargs, dimg_like
views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "two_hands"
dev = torch.device("cuda:0")
sc = make_scene(cfg, n_views=views)
s = sc.to(dev); kw, bl = scene_kwargs(s); cams = sc.cams().to(dev)
L = _lib.lib()
dimg = dimg_like(views, sc.H, sc.W).to(dev)
TUS = ("render", "pre", "bin")
for rep in range(3):
    for tu in TUS: getattr(L, f"gh_exp_wg_clear_{tu}")()
    torch.cuda.synchronize()
    img, radii, ctx = raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=True, expect_backward=True, **kw, **bl)
    g = raster_backward(ctx, dimg); torch.cuda.synchronize()
rows = []
for tu in TUS:
    buf = np.zeros((1 << 19, 4), dtype=np.uint32); n = C.c_uint32(0)
    assert getattr(L, f"gh_exp_wg_read_{tu}")(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes), C.byref(n)) == 0
    rows.append(buf[:min(n.value, 1 << 19)])
a = np.concatenate(rows)
names = {1: ("gh_render_fwd_kernel", 8), 2: ("gh_preprocess_fwd_kernel", 7), 3: ("gh_preprocess_bwd_kernel", 4), 4: ("gh_emit_kernel", 8), 5: ("gh_ranges_kernel", 8)}
def makespan(d, order, slots):
    h = [0.0] * slots; heapq.heapify(h); end = 0.0
    for i in order:
        t = heapq.heappop(h) + d[i]; end = max(end, t); heapq.heappush(h, t)
    return end
print(f"{views} views {cfg} (workgroups per CU assumed in brackets; cycles at 2.4 GHz)")
for kid, (name, per_cu) in names.items():
    r = a[a[:, 2] == kid]
    if len(r) == 0: continue
    d = r[:, 0].astype(np.float64); blk = r[:, 1]
    o = np.argsort(blk, kind="stable")
    slots = 256 * per_cu
    t_disp, t_lpt = makespan(d, o, slots) / 2.4e3, makespan(d, np.argsort(-d), slots) / 2.4e3
    q = max(len(o) // 8, 1)
    print(f"{name:28s} [{per_cu}] {len(d):6d} workgroups: mean {d.mean():7.0f} p99 {np.percentile(d, 99):7.0f} max {d.max():7.0f} cycles | as dispatched {t_disp:6.1f} us, "
          f"longest first {t_lpt:6.1f} us, sum/slots {d.sum() / slots / 2.4e3:6.1f} us, longest {d.max() / 2.4e3:5.1f} us | mean by eighth of the dispatch order: "
          + " ".join(f"{d[o[k * q:(k + 1) * q]].mean():.0f}" for k in range(8)))
