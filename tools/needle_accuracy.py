"""Gradient accuracy of the needle / giant stress scene of tests/test_gpu_parity.py::test_culling_is_conservative_... against the
float64 dense autograd oracle (oracle/oracle_torch.py) AND the float32 C oracle: which of the two float32 implementations is
closer to the float64 result, per gradient tensor. usage: [GH_RASTER_LIB=...] python tools/needle_accuracy.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.rasterizer import raster_forward, raster_backward
from guassianhand_amd.scenes import make_scene
from tests.helpers import dimg_like, max_rel, rel_l2, scene_kwargs
from tests.test_gpu_oracle_a import _oracle_a
from oracle.oracle_c import OracleRender

dev = torch.device("cuda:0")
sc = make_scene("random1k", n_views=2, P=2000)           # the test's scene at a third of its size (the dense oracle holds P x H x W)
g = torch.Generator().manual_seed(12)
sc.scaling[:667] = torch.stack([torch.full((667,), 0.03), torch.full((667,), 3e-5), torch.full((667,), 3e-5)], 1)   # needles
sc.scaling[667:767] = 0.2
sc.scaling[767:867] = 1e-7
sc.opacity[867:1200] = (1 / 255) * (1 + 0.02 * torch.randn(333, 1, generator=g))
sc.opacity[1200:1267] = 1 / 255
sc.opacity[:333] = 0.01 + 0.02 * torch.rand(333, 1, generator=g)
k = torch.arange(1333, 1667)
sc.xyz[k, 0] = ((k % 33) * 4 - 64 + 0.5).float() / 325.0
sc.xyz[k, 1] = (((k // 33) % 33) * 4 - 64 + 0.5).float() / 325.0
sc.xyz[k, 2] = 0.0
sc.xyz[1667:1733, 0] += 0.5
blend = {k: getattr(sc, k) for k in ("color_w", "color_b", "opacity_b", "xyz_b") if getattr(sc, k, None) is not None}
dimg = dimg_like(2, sc.H, sc.W)
img_a, backward_a = _oracle_a(sc, blend)
s = sc.to(dev)
kw, bl = scene_kwargs(s)
img, _, ctx = raster_forward(sc.cams().to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
flipped = (img.double().cpu() - img_a).abs().amax(dim=1) > 1e-4
print("pixels with a float32 / float64 threshold decision:", int(flipped.sum()))
dimg = dimg * (~flipped)[:, None].float()
ga = backward_a(dimg)
gh = raster_backward(ctx, dimg.to(dev), want_means2D=False)
torch.cuda.synchronize()
ckw, cbl = scene_kwargs(sc)
orc = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=False, **ckw, **cbl)
go = orc.backward(dimg)
print(f"library: {os.environ.get('GH_RASTER_LIB', 'in-tree')}")
for kname in ("means3D", "scales", "rotations", "opacities"):
    a = ga[kname]
    h = gh[kname].double().cpu().reshape(a.shape)
    o = go[kname].double().reshape(a.shape)
    print(f"{kname:10s} vs float64 oracle: HIP max_rel {max_rel(h, a):.3e} rel_l2 {rel_l2(h, a):.3e} | C oracle (float32) max_rel {max_rel(o, a):.3e} "
          f"rel_l2 {rel_l2(o, a):.3e} | HIP vs C oracle max_rel {max_rel(h, o):.3e}")
