"""How many tile instances come from Gaussians whose 3-sigma tile rect exceeds the 64-bit hit mask (the slow recount path of
gh_ranges_kernel / own-lane path of gh_emit_kernel). usage: python tools/rect_stats.py [scene] [views]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
name = sys.argv[1] if len(sys.argv) > 1 else "two_hands"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
sc = make_scene(name, n_views=nv).to(dev)
blend = dict(xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=sc.color_b)
kw = dict(colors_precomp=sc.shs.reshape(sc.P, 3)) if sc.use_rgb else dict(shs=sc.shs, sh_degree=sc.sh_degree)
img, radii, ctx = R.raster_forward(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True, **blend, **kw)
wv = R.workspace_views(ctx)
r = wv["rect"].long() & 0xFFFFFFFF
tt = wv["tiles_touched"].long()
w, h = ((r >> 16) & 255) - (r & 255), (r >> 24) - ((r >> 8) & 255)
area = w * h
D = int(tt.sum())
for lim in (16, 32, 64, 128, 256):
    big = area > lim
    print(f"{name} {sc.H}x{sc.W} {nv} views: rect area > {lim:3d} tiles: {int((big & (tt > 0)).sum()):7d} Gaussians, {int(tt[big].sum()):8d} instances "
          f"({100.0 * int(tt[big].sum()) / D:.1f} % of D={D}), sum over them of instances x area {int((tt[big] * area[big]).sum()):,}")
print("max area", int(area.max()), "max instances of one Gaussian", int(tt.max()))
