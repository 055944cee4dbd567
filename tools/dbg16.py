import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.scenes import make_scene, ring_cameras
from guassianhand_amd.rasterizer import raster_forward, workspace_views
from oracle.oracle_c import OracleRender
dev = torch.device("cuda:0")
for (H, W) in [(16, 16), (17, 33)]:
    sc = make_scene("random1k", n_views=2, P=600)
    sc.H, sc.W = H, W
    sc.w2c, sc.K = ring_cameras(torch.zeros(3), 2, H, W, 1.3 * max(H, W))
    s = sc.to(dev)
    img, radii, ctx = raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, colors_precomp=s.shs.squeeze(1))
    wv = workspace_views(ctx)
    o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=H, W=W, colors_precomp=sc.shs.squeeze(1), debug=True)
    N = 1200
    vis = o.radii.reshape(-1) > 0
    g0, g1 = wv["g0"].cpu(), wv["g1"].cpu()
    mine = torch.stack([g0[:, 2], g0[:, 3], g1[:, 0], g1[:, 1]], 1)
    ref = o.debug["conic_opacity"].reshape(N, 4)
    bad = ((mine != ref).any(1)) & vis
    print(H, W, "n bad", int(bad.sum()), "of", int(vis.sum()))
    idx = bad.nonzero().reshape(-1)[:5]
    for i in idx.tolist():
        print(i, mine[i].tolist(), ref[i].tolist(), (mine[i].view(torch.int32) - ref[i].view(torch.int32)).tolist())
        print("   xy", g0[i, :2].tolist(), o.debug["xy"].reshape(N, 2)[i].tolist(), "depth", wv["depth"][i].item(), o.debug["depth"].reshape(N)[i].item())
    print(" cams equal:", torch.equal(s.cams().cpu(), sc.cams()))
    print(" cam diff", (s.cams().cpu() - sc.cams()).abs().max().item())
