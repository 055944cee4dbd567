"""Stage-by-stage comparison of the HIP path against the C oracle on one scene (diagnostic, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.scenes import make_scene
from guassianhand_amd.rasterizer import raster_forward, raster_backward, workspace_views
from oracle.oracle_c import OracleRender


def check(config, P, rgb, blend, n_views=1):
    sc = make_scene(config, n_views=n_views, P=P, use_rgb=rgb, blend=blend)
    dev = torch.device("cuda:0")
    s = sc.to(dev)
    cams = s.cams()
    kw = dict(colors_precomp=s.shs.squeeze(1)) if rgb else dict(shs=s.shs, sh_degree=3)
    bl = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b) if blend else {}
    img, radii, ctx = raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
    torch.cuda.synchronize()
    wv = workspace_views(ctx)
    D = int(wv["counters"][0].item())
    cpu = lambda t: None if t is None else t.cpu()
    kwc = {k: cpu(v) for k, v in kw.items() if torch.is_tensor(v)}
    if not rgb: kwc["sh_degree"] = 3
    blc = {k: cpu(v) for k, v in bl.items()}
    t0 = time.time()
    orc = OracleRender(cams.cpu(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=True, **kwc, **blc)
    t_or = time.time() - t0
    print(f"== {config} P={sc.P} rgb={rgb} blend={blend} views={n_views}: D gpu {D} oracle {orc.num_rendered} (oracle fwd {t_or:.2f}s)")
    N = n_views * sc.P
    vis = orc.radii.reshape(-1) > 0
    print(" radii equal:", torch.equal(radii.cpu(), orc.radii))
    g0 = wv["g0"].cpu(); g1 = wv["g1"].cpu(); gb = wv["gb"].cpu()
    oxy = orc.debug["xy"].reshape(N, 2); oco = orc.debug["conic_opacity"].reshape(N, 4); orgb = orc.debug["rgb"].reshape(N, 3)
    print(" xy bit-equal:", torch.equal(g0[vis, :2], oxy[vis]), " conic:", torch.equal(torch.stack([g0[vis,2], g0[vis,3], g1[vis,0]],1), oco[vis,:3]),
          " opac:", torch.equal(g1[vis,1], oco[vis,3]), " rgb:", torch.equal(torch.stack([g1[vis,2], g1[vis,3], gb[vis]],1), orgb[vis]),
          " rgb maxdiff:", (torch.stack([g1[vis,2], g1[vis,3], gb[vis]],1) - orgb[vis]).abs().max().item())
    print(" depth:", torch.equal(wv["depth"].cpu()[vis], orc.debug["depth"].reshape(N)[vis]), " rect:", torch.equal(wv["rect"].cpu()[vis], orc.debug["rect"].reshape(N)[vis]),
          " tiles:", int(wv["tiles_touched"].sum()) == orc.num_rendered)
    if D == orc.num_rendered:
        print(" sorted tile:", torch.equal(wv["sorted_tile"][:D].cpu().long(), orc.debug["sorted_keys"] >> 32), " sorted gid:", torch.equal(wv["sorted_gid"][:D].cpu(), orc.debug["sorted_gid"]),
              " ranges:", torch.equal(wv["ranges"].cpu(), orc.debug["ranges"]))
    print(" final_T bit-equal:", torch.equal(wv["final_T"].cpu(), orc.debug["final_T"]), " n_contrib:", torch.equal(wv["n_contrib"].cpu(), orc.debug["n_contrib"]))
    print(" image bit-equal:", torch.equal(img.cpu(), orc.image), " L_inf:", (img.cpu() - orc.image).abs().max().item())
    g = torch.Generator().manual_seed(5)
    dimg = torch.randn(n_views, 3, sc.H, sc.W, generator=g)
    grads = raster_backward(ctx, dimg.to(dev))
    torch.cuda.synchronize()
    grads2 = raster_backward(ctx, dimg.to(dev))
    torch.cuda.synchronize()
    t0 = time.time(); og = orc.backward(dimg); t_ob = time.time() - t0
    for k in og:
        a, b = grads[k].cpu().reshape(-1), og[k].reshape(-1)
        rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
        mx = ((a - b).abs() / (b.abs() + 1e-3 * b.abs().max() + 1e-30)).max().item()
        print(f"  grad {k:14s} rel-l2 {rel:.3e} max-rel {mx:.3e} |ref|max {b.abs().max().item():.3e} deterministic {torch.equal(grads[k], grads2[k])}")
    print(f"  (oracle bwd {t_ob:.2f}s)")


if __name__ == "__main__":
    check("random1k", 1000, True, False)
    check("random1k", 1000, False, False)
    check("random1k", 3000, True, True)
    check("random1k", 3000, False, True, n_views=2)
    check("one_hand", 20000, True, False)
    check("two_hands", None, True, True)
