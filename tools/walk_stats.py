"""How much of the tile lists the forward actually walks (two_hands, 8 views): instances behind the point where every pixel
of their tile is saturated are binned, sorted and gathered but never blended."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from guassianhand_amd import rasterizer as R, _abi, _lib
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8).to(dev)
blend = dict(xyz_b=sc.xyz_b, opacity_b=sc.opacity_b, color_w=sc.color_w, color_b=sc.color_b)
img, radii, ctx = R.raster_forward(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, sync=True,
                                   colors_precomp=sc.shs.reshape(sc.P, 3), **blend)
wv = R.workspace_views(ctx)
lay = _abi.GhLayout(); _lib.lib().gh_workspace_layout(C.byref(ctx.dims), C.byref(lay))
T = wv["ranges"].shape[0]
walk = ctx.ws[lay.tile_walk:lay.tile_walk + 4 * T].view(torch.int32).long()
rng = wv["ranges"].long()
ln = rng[:, 1] - rng[:, 0]
D = int(wv["counters"][0])
print("D", D, "sum len", int(ln.sum()), "sum walked", int(torch.minimum(walk, ln).sum()), "walked fraction", float(torch.minimum(walk, ln).sum()) / D)
nc = wv["n_contrib"].long()
print("mean n_contrib per pixel (position of last blended entry)", float(nc.float().mean()), "max", int(nc.max()))
# per Gaussian: is any of its instances inside the walked prefix of its tile?
tile_of = torch.repeat_interleave(torch.arange(T, device=dev), ln)
local = torch.arange(D, device=dev) - rng[tile_of, 0]
walked = local < walk[tile_of]
gid = wv["sorted_gid"][:D].long()
N = sc.P * 8
seen = torch.zeros(N, dtype=torch.bool, device=dev)
seen[gid[walked]] = True
has = torch.zeros(N, dtype=torch.bool, device=dev); has[gid] = True
print("(view, Gaussian) pairs with instances", int(has.sum()), "with at least one walked instance", int(seen.sum()))
