"""Forward-only loop on the default bench workload (for per-kernel timing of forward stages under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
s = make_scene("two_hands", n_views=V).to(dev)
blend = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
with torch.no_grad():
    for i in range(12):
        R.rasterize_views(s.cams().contiguous(), s.xyz, s.opacity, s.scaling, s.rotation, s.shs, H=s.H, W=s.W, use_rgb=True,
                          sync=(i == 0), **blend)
torch.cuda.synchronize()
print("ok")
