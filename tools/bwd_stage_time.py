"""Stage times (HIP events) of forward + backward on a scene with the colour mode forced: python tools/bwd_stage_time.py <scene> <views> <rgb 0|1>
(GH_RASTER_LIB selects the library build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
from tests.helpers import scene_kwargs, dimg_like
name, nv, rgb = sys.argv[1], int(sys.argv[2]), bool(int(sys.argv[3]))
dev = torch.device("cuda:0")
sc = make_scene(name, n_views=nv, use_rgb=rgb).to(dev)
kw, bl = scene_kwargs(sc)
dimg = dimg_like(nv, sc.H, sc.W).to(dev)
def step():
    img, _, ctx = R.raster_forward(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True, **kw, **bl)
    R.raster_backward(ctx, dimg, want_means2D=False)
for _ in range(3): step()
R.enable_stage_timing(True)
for _ in range(10): step()
t = R.stage_timing_summary()
print(f"{os.environ.get('GH_RASTER_LIB', 'in-tree')}: {name} {sc.H}x{sc.W} {nv} views rgb={rgb} D={R.last_num_rendered()}: " + ", ".join(f"{k} {v * 1e3:.1f} us" for k, v in t.items()))
