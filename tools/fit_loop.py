"""N eager full-size fit steps (8 views, two hands, active texels) for rocprofv3: python tools/fit_loop.py <static 0|1> [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
static = sys.argv[1] == "1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = make_scene("two_hands", n_views=nv, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
f = F.OneShotFit(gs, uv, static_geometry=static)
with torch.no_grad():
    out = f.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, f.blend_values())
gt_rgb, gt_mask = (out["comp_rgb"] * 0.9).contiguous(), out["comp_mask"].mean(-1).contiguous()
args = (sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask)
for i in range(n):
    f.step(*args, sync=(i == 0))
torch.cuda.synchronize()
R.check_overflow()
