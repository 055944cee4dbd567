"""How often, and where, the speculative occlusion bound (DepthBoundCache) misses on a moving-geometry loop of the hand scenes: per-step misses,
the tiles concerned and what margin / slack would have held (the measurements behind the virtual walk and the neighbourhood erosion of DESIGN.md 5e)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
V = 8
s = make_scene("two_hands", n_views=V).to(dev)
cams = s.cams()
kw = dict(H=s.H, W=s.W, colors_precomp=s.shs.squeeze(1), xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
g = torch.Generator().manual_seed(3)
cache = R.DepthBoundCache(margin=2e-3, slack=32, min_pixels=0)
xyz = s.xyz.clone()
R.raster_forward(cams, xyz, s.opacity, s.scaling, s.rotation, sync=True, depth_bound=cache, **kw)
gx, gy = (s.W + 15) // 16, (s.H + 15) // 16
for step in range(4):
    xyz = xyz + 1e-4 * torch.randn(xyz.shape, generator=g).to(dev)
    seen_prev = cache.bufs[cache.cur].clone()
    img_b, _, ctx_b = R.raster_forward(cams, xyz, s.opacity, s.scaling, s.rotation, sync=False, depth_bound=cache, **kw)
    try:
        R.check_overflow()
        print("step", step, "hit")
        continue
    except R.GhDepthBoundMiss:
        pass
    lay_views = None
    img_u, _, ctx_u = R.raster_forward(cams, xyz, s.opacity, s.scaling, s.rotation, sync=True, **kw)
    wv = R.workspace_views(ctx_u)
    nan = torch.isnan(img_b).any(1)      # (V,H,W)
    idx = torch.nonzero(nan)
    print("step", step, "MISS: nan pixels", idx.shape[0])
    fT = wv["final_T"]; nc = wv["n_contrib"]
    # effective bound of the bounded call
    import ctypes as C
    from guassianhand_amd import _abi, _lib
    lay = _abi.GhLayout(); _lib.lib().gh_workspace_layout(C.byref(ctx_b.dims), C.byref(lay))
    T = V * gx * gy
    tb = ctx_b.ws[lay.tile_bound:lay.tile_bound + 4 * T].view(torch.float32)
    rng = wv["ranges"].long(); gid = wv["sorted_gid"].long(); depth = wv["depth"]
    for (v, y, x) in idx[:12].tolist():
        tile = v * gx * gy + (y // 16) * gx + (x // 16)
        r0, r1 = rng[tile].tolist()
        last = int(nc[v, y, x])
        dlast = float(depth[gid[r0 + last - 1]]) if last > 0 else float('nan')
        dend = float(depth[gid[r1 - 1]]) if r1 > r0 else float('nan')
        sp = seen_prev[tile]
        print(f"  v{v} px({x},{y}) tile({x//16},{y//16}) unbounded: final_T {float(fT[v,y,x]):.3e} n_contrib {last} of {r1-r0} depth(last blended) {dlast:.4f} list end depth {dend:.4f} | eff bound {float(tb[tile]):.4f} reported (depth,mask) ({float(sp[0]):.4f}, {int(sp[1].view(torch.int32)):#06x})")
    cache.clear()
    R.raster_forward(cams, xyz, s.opacity, s.scaling, s.rotation, sync=True, depth_bound=cache, **kw)
