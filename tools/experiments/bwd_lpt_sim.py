"""How much of the render backward is its work-list ORDER? Per-workgroup cycle counters (library built with -DGH_EXP_BWD_TIME:
bash tools/abl_build.sh bwdtime -DGH_EXP_BWD_TIME; GH_RASTER_LIB=tools/abl/bwdtime.so) of one backward over the bench's default step,
then greedy list scheduling of the measured durations over the chip's wave slots: in the order the kernel took them, in LPT order
(longest first), and the two lower bounds (sum / slots, longest item). usage: bwd_lpt_sim.py [views] [config]"""
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
dumps = []
for rep in range(4):
    img, radii, ctx = raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=True, expect_backward=True, **kw, **bl)
    L.gh_exp_clear_bwd_times(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g = raster_backward(ctx, dimg); e1.record(); torch.cuda.synchronize()
    buf = np.zeros((1 << 18, 4), dtype=np.uint32)
    assert L.gh_exp_read_bwd_times(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
    dumps.append(buf[buf[:, 0] > 0].copy())
if len(sys.argv) > 3:
    np.savez_compressed(sys.argv[3], *dumps)
used = buf[:, 0] > 0
d = buf[used, 0].astype(np.float64); slot = np.nonzero(used)[0]
print(f"{views} views {cfg}: backward (all its kernels, events) {e0.elapsed_time(e1) * 1e3:.0f} us; {used.sum()} workgroups did work, of {slot.max() + 1} launched")
print(f"durations (cycles): mean {d.mean():.0f}, median {np.median(d):.0f}, p90 {np.percentile(d, 90):.0f}, p99 {np.percentile(d, 99):.0f}, max {d.max():.0f}; sum {d.sum() / 1e6:.1f} M")
def makespan(order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + d[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end
for slots in (256 * 4 * 5,):
    act = makespan(np.argsort(slot), slots)             # dispatch order = workgroup index
    lpt = makespan(np.argsort(-d), slots)
    rnd = makespan(np.random.default_rng(0).permutation(len(d)), slots)
    print(f"{slots} wave slots: as dispatched {act / 2.4e3:.1f} us | LPT {lpt / 2.4e3:.1f} us | random {rnd / 2.4e3:.1f} us | bounds: sum/slots {d.sum() / slots / 2.4e3:.1f} us, longest {d.max() / 2.4e3:.1f} us  (cycles at 2.4 GHz)")
# where the long ones sit in the dispatch order
o = np.argsort(slot)
q = len(o) // 10
print("mean duration by tenth of the dispatch order:", " ".join(f"{d[o[k * q:(k + 1) * q]].mean():.0f}" for k in range(10)))
top = np.argsort(-d)[:20]
print("the 20 longest: dispatch position (of %d) / cycles:" % len(d), " ".join(f"{np.searchsorted(np.sort(slot), slot[i])}/{int(d[i])}" for i in top))
