"""A loop whose Gaussians move a little every step (positions + N(0, sigma) per step: what the one-shot fit's network-side
trainables do, infer_one_shot.py:340-343) with and without the speculative occlusion bound (rasterizer.DepthBoundCache):
instances D, per-stage GPU time (HIP events on the launch stream), their sum, and how often the bound missed.
usage: python tools/moving_geometry.py [views] [sigma_m] [margin] [config]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.loss import rendered_l1_loss
from guassianhand_amd.scenes import make_scene, perturbed_target_xyz

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
margin = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-3
config = sys.argv[4] if len(sys.argv) > 4 else "two_hands"
slack = int(sys.argv[5]) if len(sys.argv) > 5 else 8
refresh = int(sys.argv[7]) if len(sys.argv) > 7 else 4
mode = sys.argv[6] if len(sys.argv) > 6 else "both"          # "none" / "bound": one of the two loops only (for rocprofv3 --stats)
dev = torch.device("cuda:0")
sc = make_scene(config, n_views=V)
s = sc.to(dev)
cams = s.cams()
blend = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
with torch.no_grad():
    gt, _ = R.rasterize_views(cams, perturbed_target_xyz(sc).to(dev), s.opacity, s.scaling, s.rotation, s.shs, H=s.H, W=s.W, use_rgb=s.use_rgb,
                              sh_degree=s.sh_degree, **blend)
K = 24
g = torch.Generator().manual_seed(3)
walk = torch.cumsum(sigma * torch.randn(K + 4, *sc.xyz.shape, generator=g), 0)       # a random walk: every step moves on from the last
xyzs = [(sc.xyz + walk[k]).to(dev).requires_grad_(True) for k in range(K + 4)]
params = {k: getattr(s, k).clone().requires_grad_(True) for k in ("opacity", "scaling", "rotation", "shs", "opacity_b", "color_w", "color_b")}
seed = torch.ones((), device=dev)


def run(cache):
    def step(k, sync):
        loss, _, _ = rendered_l1_loss(cams, xyzs[k], params["opacity"], params["scaling"], params["rotation"], params["shs"], gt, H=s.H, W=s.W,
                                      use_rgb=s.use_rgb, sh_degree=s.sh_degree, sync=sync, xyz_b=s.xyz_b, opacity_b=params["opacity_b"],
                                      color_w=params["color_w"], color_b=params["color_b"], depth_bound=cache)
        loss.backward(seed)
        return loss
    Ds = []
    for k in range(4):                    # sync: D of a step (and a transparent re-run on a miss)
        step(k, True)
        Ds.append(R.last_num_rendered())
    torch.cuda.synchronize()
    # sync-free steps: the host runs ahead, the events bracket GPU time only (as bench.py's stage leg does)
    R.enable_stage_timing(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    missed = 0
    for k in range(4, K + 4):
        step(k, False)
    e1.record()
    st = R.stage_timing_summary()
    R.enable_stage_timing(False)
    try:
        R.check_overflow()
    except R.GhOverflowError:
        missed = 1
    return Ds[-1], st, e0.elapsed_time(e1) / K, missed


for name, cache in (("no bound", None), ("occlusion bound", R.DepthBoundCache(margin=margin, slack=slack, refresh_every=refresh, min_pixels=0))):
    if (mode == "none" and cache is not None) or (mode == "bound" and cache is None):
        continue
    D, st, wall, missed = run(cache)
    tot = sum(st.values())
    print(f"{config} x {V} views, sigma {sigma * 1e3:.2f} mm / step, margin {margin:g} slack {slack} refresh {refresh} | {name:16s}: D {D / 1e6:6.3f} M  " +
          "  ".join(f"{k} {v * 1e3:6.1f}" for k, v in st.items()) + f"  | stages {tot * 1e3:6.1f} us, step {wall * 1e3:6.1f} us (eager, with events)" +
          (f"  misses {cache.misses} of {cache.bounded_calls} bounded calls" if cache is not None else ""))
