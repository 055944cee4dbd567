"""Full-size one-shot fit step (8 views, two hands P = 98,562, 1024x2048 maps, active texels): static geometry (tile lists
built once, gh_forward_refresh per step) against the full path, eager and as a captured HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = make_scene("two_hands", n_views=nv, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)


def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for static in (False, True):
    f = F.OneShotFit(gs, uv, static_geometry=static)
    with torch.no_grad():
        out = f.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, f.blend_values())
    gt_rgb, gt_mask = (out["comp_rgb"] * 0.9).contiguous(), out["comp_mask"].mean(-1).contiguous()
    args = (sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask)
    for i in range(3): f.step(*args, sync=(i == 0))
    D = R.last_num_rendered()
    eager = t(lambda: f.step(*args, sync=False))
    R.check_overflow()
    l_eager = float(f.step(*args, sync=False))
    cap = f.captured(*args)
    l_cap0 = float(cap.replay())
    graph = t(lambda: cap.replay())
    cap.check()
    loss = float(cap.replay())
    print(f"   losses: after the eager loop {l_eager:.6f}, first replay {l_cap0:.6f}, last replay {loss:.6f}")
    print(f"fit step, {nv} views, P={sc.P}, static_geometry={static}: eager {eager:.3f} ms, captured graph {graph:.3f} ms; "
          f"instances D={D}; loss after the timed steps {loss:.6f}" +
          (f"; cache builds/hits {f._geom_cache.builds}/{f._geom_cache.hits}" if static else ""))
    del f, cap
    torch.cuda.empty_cache()
