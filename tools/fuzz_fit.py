"""Sequence fuzz of the one-shot fit's three execution modes (fit.OneShotFit / fit.CapturedFitStep).

Three fits of the same small problem are driven through the same random schedule:
    A  static geometry (tile lists built once, gh_forward_refresh afterwards), eager steps
    B  static geometry, every step a replay of the captured HIP graph (re-captured at learning-rate milestones, eager + re-capture
       when the lists it was captured on were dropped)
    C  full path every step (static_geometry=False)
    D  full path through a DepthBoundCache (occlusion_bound=True: the policy for Gaussians that move every step) — must take C's steps
Events: steps, epoch ends (learning-rate milestones), update_gaussians() with Gaussians that moved a little, invalidate_geometry(),
GeometryCache.clear_all() (what an overflow anywhere in the process does), clear_workspace_pool(), unrelated renders of other sizes in
between (they compete for pooled workspaces). After every event batch: A and B hold the SAME bits (parameters, both Adam moments,
losses); C agrees with A to float32 accumulation (the static lists hold more instances, so sums run in another order).
usage: fuzz_fit.py [n_iterations] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F
from guassianhand_amd import rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
from tests.helpers import tiny_fit_problem

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--only=")]
rnd = random.Random(seed)
dev = torch.device("cuda:0")
stats = dict(steps=0, replays=0, epoch_ends=0, moves=0, invalidations=0, clear_alls=0, pool_clears=0, other_renders=0, recaptures=0)
bad = []
worst_moment = [0.0]


def one(it):
    rnd.seed(seed * 1000003 + it)
    P = rnd.choice([200, 600])
    hw = (rnd.choice([48, 64, 80]), rnd.choice([48, 64, 80]))
    use_rgb = rnd.random() < 0.6
    pb = tiny_fit_problem(P=P, n_views=4, hw=hw, device=dev, seed=it)
    g = torch.Generator().manual_seed(it)
    gt_rgb = torch.rand(4, hw[0], hw[1], 3, generator=g).to(dev)
    gt_mask = (torch.rand(4, hw[0], hw[1], generator=g) > 0.5).float().to(dev)
    args = (pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
    gs = pb["gs"]
    if not use_rgb:
        shs = torch.cat([gs.shs, 0.1 * torch.randn(gs.shs.shape[0], 15, 3, generator=g).to(dev)], 1)
        gs = GaussianModel(gs.xyz, gs.opacity, gs.rotation, gs.scaling, shs)
    mk = lambda static: F.OneShotFit(gs, pb["uv"], map_hw=pb["map_hw"], use_rgb=use_rgb, static_geometry=static)
    A, B, C = mk(True), mk(True), mk(False)
    # D: the policy for Gaussians that move every step — full forwards through a DepthBoundCache (speculative occlusion bound, verified by
    # the forward, re-run on a miss); made to apply to these small renders too
    D = F.OneShotFit(gs, pb["uv"], map_hw=pb["map_hw"], use_rgb=use_rgb, static_geometry=False, occlusion_bound=True)
    D._geom_cache.min_pixels, D._geom_cache.refresh_every = 0, rnd.choice([1, 2, 4])
    ld = []
    trace = []
    tag = lambda: f"it {it} P={P} {hw} {'rgb' if use_rgb else 'sh3'}: " + " > ".join(trace[-14:])
    la, lb, lc = [], [], []
    # B's construction runs two regular steps: the others take them too
    for f, ls in ((A, la), (C, lc), (D, ld)):
        for i in range(2):
            ls.append(float(f.step(*args, sync=(i == 0))))
    cap = B.captured(*args)
    trace.append("construct(2 steps)")
    stats["steps"] += 2
    other = make_scene("random1k", n_views=1, P=500).to(dev)

    def compare():
        for k in A._adam:
            a, b, c = A._adam[k], B._adam[k], C._adam[k]
            for name in ("param", "exp_avg", "exp_avg_sq"):
                if not torch.equal(getattr(a, name), getattr(b, name)):
                    x, y, z = getattr(a, name), getattr(b, name), getattr(c, name)
                    raise AssertionError(tag() + f" (eager static vs captured: {k}.{name}: {int((x != y).sum())} of {x.numel()} elements differ, max |A-B| "
                                         f"{float((x - y).abs().max()):.3g}; max |A-C| {float((x - z).abs().max()):.3g}, max |B-C| {float((y - z).abs().max()):.3g}; "
                                         f"losses A {la[-4:]} B {lb[-4:]} C {lc[-4:]})")
            assert int(a.step_state[0]) == int(b.step_state[0]) == int(c.step_state[0]), tag() + " (applied step counts)"
            # static vs full path: the gradients agree to float32 accumulation; Adam turns a gradient that IS accumulation noise
            # (a texel no pixel constrains: g ~ 1e-12) into a step of +-lr, and the L1 regulariser's sign(color_b) flips on a texel that
            # hovers at +-1e-9: single parameters / moments may differ by a few lr while the losses and nearly all parameters agree
            m_scale = float(a.exp_avg.abs().max()) + 1e-30
            dm = float((a.exp_avg - c.exp_avg).abs().max())
            worst_moment[0] = max(worst_moment[0], dm / m_scale)
            diff = (a.param - c.param).abs().reshape(-1)
            q = float(torch.quantile(diff[:1 << 22].float(), 0.99)) if diff.numel() else 0.0
            assert q <= 1e-3, tag() + f" (static vs full path: {k}, 99th percentile of |diff| {q:.3g}, max {float(diff.max()):.3g})"
        assert la[2:] == lb, tag() + f" (losses eager static vs captured: {la[-3:]} vs {lb[-3:]})"
        for x, y in zip(lc, ld):                                  # the occlusion bound is exact: the full path's losses
            assert abs(x - y) <= 2e-6 * max(1.0, abs(x)), tag() + f" (loss full path {x} vs occlusion-bound fit {y})"
        for k in C._adam:
            dq = (C._adam[k].param - D._adam[k].param).abs().reshape(-1)
            qd = float(torch.quantile(dq[:1 << 22].float(), 0.99)) if dq.numel() else 0.0
            assert qd <= 1e-3 and int(C._adam[k].step_state[0]) == int(D._adam[k].step_state[0]), tag() + f" (full path vs occlusion-bound fit: {k}, 99th percentile {qd:.3g})"
        for x, y in zip(la, lc):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), tag() + f" (loss static {x} vs full {y})"

    for _ in range(rnd.randint(6, 16)):
        ev = rnd.choice(["steps"] * 5 + ["epochs", "move", "invalidate", "clear_all", "pool", "other"])
        if ev == "steps":
            k = rnd.randint(1, 4)
            trace.append(f"steps({k})")
            for _i in range(k):
                la.append(float(A.step(*args, sync=rnd.random() < 0.3)))
                lc.append(float(C.step(*args, sync=rnd.random() < 0.3)))
                ld.append(float(D.step(*args, sync=True)))
                g0 = cap.graph
                lb.append(float(cap.replay()))
                stats["recaptures"] += 0 if cap.graph is g0 else 1
                stats["steps"] += 1
                stats["replays"] += 1
        elif ev == "epochs":
            k = rnd.randint(1, 6)
            trace.append(f"end_epoch x{k}")
            stats["epoch_ends"] += k
            for f in (A, B, C, D):
                for _i in range(k):
                    f.end_epoch()
        elif ev == "move":
            trace.append("update_gaussians")
            stats["moves"] += 1
            cur = A.gs
            d = rnd.choice([0.002, 0.002, 0.05]) * torch.randn(cur.xyz.shape, generator=g).to(dev)      # (5 cm: the occlusion bound of fit D misses)
            new = GaussianModel(cur.xyz + d, cur.opacity, cur.rotation, cur.scaling, cur.shs)
            for f in (A, B, C, D):
                f.update_gaussians(new)
        elif ev == "invalidate":
            trace.append("invalidate_geometry")
            stats["invalidations"] += 1
            for f in (A, B, C, D):
                f.invalidate_geometry()
        elif ev == "clear_all":
            trace.append("GeometryCache.clear_all")
            stats["clear_alls"] += 1
            R.GeometryCache.clear_all()
        elif ev == "pool":
            trace.append("clear_workspace_pool")
            stats["pool_clears"] += 1
            R.clear_workspace_pool()
        else:
            hh, ww = rnd.randint(30, 120), rnd.randint(30, 120)
            trace.append(f"other_render({hh}x{ww})")
            stats["other_renders"] += 1
            other.H, other.W = hh, ww
            R.raster_forward(other.cams(), other.xyz, other.opacity, other.scaling, other.rotation, H=hh, W=ww,
                             colors_precomp=other.shs.squeeze(1), sync=rnd.random() < 0.5)
        compare()
    cap.check()
    R.check_overflow()
    stats["bounded_calls"] = stats.get("bounded_calls", 0) + D._geom_cache.bounded_calls
    stats["bound_misses"] = stats.get("bound_misses", 0) + D._geom_cache.misses


t0 = time.time()
for it in (only or range(n_iter)):
    try:
        one(it)
    except AssertionError as e:
        bad.append(str(e)[:700])
        print("MISMATCH", bad[-1], flush=True)
    except Exception as e:
        bad.append(f"it {it}: {type(e).__name__}: {e}"[:700])
        print("ERROR", bad[-1], flush=True)
    if (it + 1) % 10 == 0:
        print(f"{it + 1} iterations, {len(bad)} findings, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"fit sequence fuzz: {n_iter} iterations (seed {seed}): {stats}; {len(bad)} findings; static vs full path: largest first-moment difference {worst_moment[0]:.2e} of the moment's scale")
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
