"""Time a full-size one-shot fit step (8 views, two hands, 1024x2048 maps) in both map modes and its parts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
from guassianhand_amd.uvmap import uv_gather_backward
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)


def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for active in (True, False):
    f = F.OneShotFit(gs, uv, active_texels=active)
    with torch.no_grad():
        out = f.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, f.blend_values())
    gt_rgb, gt_mask = (out["comp_rgb"] * 0.9).clone(), out["comp_mask"].mean(-1).clone()
    for i in range(3): f.step(sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask, sync=(i == 0))
    ms = t(lambda: f.step(sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask, sync=False))
    mode = f"active texels U={f.texels.U}" if active else "dense maps + torch.optim.Adam"
    print(f"fit step [{mode}]: {ms:.3f} ms (8 views, P={sc.P}, maps 48x1024x2048)")
    print("  blend_values (lookup fwd): %.3f ms" % t(lambda: f.blend_values()))
    if active:
        go = torch.ones(sc.P, f.cb_channels, device=dev)
        print(f"  gather backward ({f.cb_channels} ch): %.3f ms" % t(lambda: uv_gather_backward(go, f.texels, f._adam["color_b"].grad)))
        f._adam["color_b"].grad.zero_()
        print("  fused regulariser + Adam (3 tensors): %.3f ms" % t(lambda: [a.step() for a in f._adam.values()]))
    else:
        def gs_bwd():
            b = f.blend_values(); (b["color_b"].sum() + b["opacity_b"].sum()).backward()
        print("  lookup fwd+bwd: %.3f ms" % t(gs_bwd))
        print("  regulariser fwd+bwd: %.3f ms" % t(lambda: f.regulariser().backward()))
        f.opt.zero_grad(); f.regulariser().backward()
        print("  Adam step: %.3f ms" % t(lambda: f.opt.step()))
    del f
    torch.cuda.empty_cache()
R.check_overflow()
