"""Gradient accuracy of the HIP backward vs the C oracle (double accumulation) on the headline workload, for two kinds of
upstream gradient: random normal (what tests/test_gpu_parity.py::compare uses) and the sign-type gradient of an L1 loss
(what the fit loop produces: heavy cancellation in the colour sums). Usage: python tools/bwd_accuracy.py [config] [views]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.rasterizer import raster_backward, raster_forward
from guassianhand_amd.scenes import make_scene, perturbed_target_xyz
from oracle.oracle_c import OracleRender
from tests.helpers import max_rel, rel_l2, scene_kwargs

cfg = sys.argv[1] if len(sys.argv) > 1 else "two_hands"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
sc = make_scene(cfg, n_views=nv)
s = sc.to(dev)
kw, bl = scene_kwargs(s)
cams = sc.cams().to(dev)
img, _, ctx = raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
kwc, blc = scene_kwargs(sc)
orc = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, **kwc, **blc)
assert torch.equal(img.cpu(), orc.image)
gt, _, _ = raster_forward(cams, perturbed_target_xyz(sc).to(dev), s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
g = torch.Generator().manual_seed(5)
ups = {"normal": torch.randn(nv, 3, sc.H, sc.W, generator=g), "l1_sign": (torch.sign(img - gt) / img.numel()).cpu()}
for name, d in ups.items():
    gg = raster_backward(ctx, d.to(dev))
    og = orc.backward(d)
    print(name, {k: (f"{rel_l2(gg[k].cpu(), og[k]):.2e}", f"{max_rel(gg[k].cpu(), og[k]):.2e}") for k in og})
