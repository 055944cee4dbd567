#!/usr/bin/env python
"""bench.py — forward+backward renders/s of the Gaussian-splatting hot path on MI355X.

Metric (BASELINE.json): fwd+bwd renders/sec @512x334, ~100k Gaussians, with the dominant kernel priced
against the HBM roofline. One *render* = one image forward + its backward under L = mean|img - gt|
(SURVEY.md §8d). One *step* = one pass of the hot path over one batch: `--views-per-step` cameras of the
ring (the 8 novel views of the one-shot fit loop, BASELINE configs[3]) rendered in one view-batched launch
sequence per rank, gradients w.r.t. all Gaussian attributes and the blend parameters, and (N>1) one RCCL
all-reduce of the scalar loss (views are independent; `--allreduce-grads` adds the fused gradient block, as the
sharded fit loop needs). Workload = BASELINE configs[2]: two interacting hands,
P = 98,562 Gaussians, interaction-aware attribute blend on, RGB colours, 512x334.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. Inputs are resident in HBM before the timed region starts. The timed steps replay one
captured step (forward + loss + backward) from a HIP graph (`--no-graph`: kernel-by-kernel enqueue); collectives stay
outside the graph; stage times for the roofline object come from eager steps with HIP events right after the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


WARM_MS = 40.0        # minimum duration of the untimed graph-replay warm-up (see main)


def algorithmic_bytes(P, NV, H, W, D, C=12, M=0):
    """SURVEY.md §8(d) per-stage algorithmic HBM bytes for one launch sequence over NV views, D instances."""
    T = NV * ((W + 15) // 16) * ((H + 15) // 16)
    tb = 1
    while (1 << tb) < T:
        tb += 1
    n_pass = 4 + (tb + 7) // 8
    N = P * NV
    return {
        "preprocess_fwd": N * (44 + C) + N * 48,
        "binning": N * 8 + N * 20 + D * 12 + D * 24 * n_pass + D * 8 + T * 8,
        "render_fwd": D * 40 + T * 8 + NV * H * W * 20,
        "render_bwd": NV * H * W * 20 + T * 8 + D * 40 + D * 36,
        "preprocess_bwd": N * (36 + 44 + C + 4) + N * (56 + C),
    }


def source_hash() -> str:
    """sha256 over the HIP sources + the C-ABI header: ties committed profile data to the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "guassianhand_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(src, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "gh_raster.h"), "rb").read())
    return h.hexdigest()[:16]


PROFILE_TAG = "r6"        # profiles/<tag>_pmc_*.json: the committed counter passes this build's bench lines quote


def workload_key(args, V: int):
    """Which committed counter summary (tools/refresh_profiles.sh) belongs to this run's workload, or None."""
    if args.scaling != "weak":
        return None
    if args.config == "two_hands" and V == 8 and not args.pose_batch:
        return "default"
    if args.config == "two_hands_hd" and V == 8 and not args.pose_batch:
        return "hd_sh3"
    if args.config == "two_hands_hd" and V == 32 and args.pose_batch:
        return "hd_sh3_pose32"
    return None


def pmc_profile(kind: str, kernel: str, args, V: int):
    """Per-kernel entry of a committed rocprofv3 counter summary (profiles/<tag>_pmc_<kind>[_<workload>].json), or (None, why).
    The counters come from separate profiling runs (`tools/refresh_profiles.sh`), NOT from this run: the entry is returned
    with its source tag and only when it was collected on this workload AND on this build (source_hash)."""
    w = workload_key(args, V)
    if w is None:
        return None, "no offline counter pass exists for this workload (default, two_hands_hd x 8 views, two_hands_hd pose batch 32)"
    name = f"{PROFILE_TAG}_pmc_{kind}" + ("" if w == "default" else "_" + w)
    path = os.path.join(ROOT, "profiles", name + ".json")
    if not os.path.exists(path):
        return None, f"profiles/{name}.json missing"
    try:
        doc = json.load(open(path))
        ent = doc["kernels"][kernel]
    except (KeyError, ValueError):
        return None, f"no entry for {kernel} in profiles/{name}.json"
    if doc.get("source_hash") != source_hash():
        return None, f"profiles/{name}.json is stale (collected on source {doc.get('source_hash')}, this build is {source_hash()})"
    return ent, f"profiles/{name}.json (offline rocprofv3 --pmc passes, source {doc['source_hash']})"


def effective_cpus() -> int:
    """CPUs this process can actually use: the affinity mask and the cgroup CPU quota, not the host's core count (a container
    on the GPU box sees 256 host CPUs and owns a share of 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(scene, seconds: float, torch_reference: bool = False):
    """The CPU restatement on the host cores: a reported baseline only. Default: the C oracle (oracle/gh_oracle.c, OpenMP over
    tiles). torch_reference (BASELINE configs[0], `--config random1k`): the literal "PyTorch CPU autograd reference" — the dense
    pixel x Gaussian oracle (oracle/oracle_torch.py), forward + autograd backward, torch threads = all cores."""
    s = scene
    cams = s.cams()[:1]
    g = torch.Generator().manual_seed(11)
    dimg = torch.randn(1, 3, s.H, s.W, generator=g) / (3 * s.H * s.W)
    if torch_reference:
        from oracle import oracle_torch as OT
        torch.set_num_threads(effective_cpus())          # (not the host's core count: oversubscribing a container's share is slower)
        c = cams[0]
        n, t0 = 0, time.perf_counter()
        while True:
            leaves = [t.clone().requires_grad_(True) for t in (s.xyz, s.opacity, s.scaling, s.rotation, s.shs)]
            bl = {k: getattr(s, k).clone().requires_grad_(True) for k in ("color_w", "xyz_b", "color_b", "opacity_b") if getattr(s, k) is not None}
            means, opac, cols, sh = OT.blend_attributes(leaves[0], leaves[1], leaves[4], use_rgb=s.use_rgb, **bl)
            kw = dict(colors_precomp=cols) if s.use_rgb else dict(shs=sh, sh_degree=s.sh_degree)
            img, _ = OT.rasterize_dense(means, opac, leaves[2], leaves[3], viewmatrix=c[:16].reshape(4, 4), projmatrix=c[16:32].reshape(4, 4),
                                        campos=c[32:35], tanfovx=float(c[35]), tanfovy=float(c[36]), bg=c[37:40], H=s.H, W=s.W, **kw)
            (img * dimg[0]).sum().backward()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds and n >= 2:
                break
        from oracle import oracle_c
        return {"value": n / dt, "unit": "renders/s", "cores": torch.get_num_threads(), "kind": "port",
                "threads": torch.get_num_threads(), "host_cpus": os.cpu_count(), "effective_cpus": effective_cpus(),
                "cpu_model": oracle_c.cpu_model(), "serial_fraction": None,
                "sample": f"{n} fwd+bwd renders of view 0 of the same workload ({dt:.1f} s), oracle/oracle_torch.py: dense PyTorch CPU "
                          "autograd (BASELINE configs[0]'s reference)"}
    from oracle import oracle_c
    kw = dict(colors_precomp=s.shs.squeeze(1)) if s.use_rgb else dict(shs=s.shs, sh_degree=s.sh_degree)

    def renders(budget_s, min_n):
        oracle_c.timing(reset=True)
        n, t0 = 0, time.perf_counter()
        while True:
            r = oracle_c.OracleRender(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, xyz_b=s.xyz_b,
                                      opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b, **kw)
            r.backward(dimg)
            r.close()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s and n >= min_n:
                break
        t_lib, t_ser = oracle_c.timing()
        return n, dt, t_lib, t_ser

    # Baseline mode: every stage under OpenMP (the checker's default keeps emit / sort / chain rule on one thread). The thread count
    # is the one that is FASTEST on this host: a container's CPU share is usually far below the host's core count, and the default
    # (all host cores) oversubscribes it — measured on the GPU box: 128 threads 2.5 renders/s, 32 threads 8.1
    # (profiles/r4_cpu_baseline_modes.txt). `value` counts library time (the Python wrapper's allocations are not the algorithm).
    default_threads = oracle_c.num_threads()
    oracle_c.set_parallel(True)
    try:
        sweep = {}
        cands = sorted({t for t in (4, 8, 16, 32, 64, 128, default_threads, effective_cpus()) if 1 <= t <= max(default_threads, os.cpu_count() or 1)})
        per = max(0.8, min(2.0, 0.35 * seconds / max(1, len(cands))))
        for t in cands:
            oracle_c.set_num_threads(t)
            n, dt, t_lib, _ = renders(per, 1)
            sweep[t] = n / t_lib
        best = max(sweep, key=sweep.get)
        oracle_c.set_num_threads(best)
        n, dt, t_lib, t_ser = renders(max(2.0, 0.65 * seconds), 3)
    finally:
        oracle_c.set_parallel(False)
        oracle_c.set_num_threads(default_threads)
    return {"value": n / t_lib, "unit": "renders/s", "cores": best, "kind": "port",
            "threads": best, "host_cpus": os.cpu_count(), "effective_cpus": effective_cpus(), "cpu_model": oracle_c.cpu_model(),
            "serial_fraction": t_ser / t_lib if t_lib > 0 else None,
            "value_by_wall_clock": n / dt, "thread_sweep_renders_per_s": {str(k): round(v, 2) for k, v in sweep.items()},
            "sample": f"{n} fwd+bwd renders of view 0 of the same workload ({dt:.1f} s), oracle/gh_oracle.c in its baseline mode: "
                      "projection, instance emit + per-tile sort, both render walks and the chain rule under OpenMP, at the thread "
                      "count that was fastest in a short sweep on this host (a container's CPU share is below the host's core count); "
                      "value = renders per second of library time; serial_fraction = library time on one thread / library time"}


def two_call_cost(s, view_ids, n_iter: int = 30):
    """per-view cost of the REFERENCE's protocol through the drop-in, as a maintainer who changes nothing gets it — per view
    the attribute blend in torch (renderer_one_shot.py:298-334), then the RGB call and the mask call of :338-346 / :372-379 through
    GaussianRasterizer; ONE loss over the step's views (L1 + mask MSE) and ONE backward, gradients w.r.t. every Gaussian attribute
    and blend parameter (forward_single_batch loops the views of a batch item, :494-503; the loss and its backward come once per
    step). Returns milliseconds per view of a step over `view_ids`; host-bound.
    ONE harness for every caller (this file's JSON line, tools/two_call_cost.py): rounds 3-5 had two — the bench's ran a backward of
    its own after every view over nine leaves, the tool's one backward per eight views over a single leaf — and their figures for
    "the same protocol" lay 56 % apart (profiles/r6_two_call_reconcile.txt)."""
    import math
    from guassianhand_amd.camera import Camera
    from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    settings = []
    for v in view_ids:
        cam = Camera.from_w2c(s.w2c[v], s.K[v], s.H, s.W)
        tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        mk = lambda bg, deg, cam=cam, tfx=tfx, tfy=tfy: GaussianRasterizationSettings(
            image_height=s.H, image_width=s.W, tanfovx=tfx, tanfovy=tfy, bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform.float(), sh_degree=deg, campos=cam.camera_center, prefiltered=False, debug=False)
        settings.append(mk)
    leaves = {k: getattr(s, k).clone().requires_grad_(True) for k in ("xyz", "opacity", "scaling", "rotation", "shs", "color_w", "color_b", "opacity_b")
              if getattr(s, k) is not None}
    zero = torch.zeros(3, device=s.xyz.device)
    gt = torch.rand(3, s.H, s.W, device=s.xyz.device)

    def one():
        for p in leaves.values():
            p.grad = None
        loss = 0
        for mk in settings:
            means, op = leaves["xyz"], leaves["opacity"]
            if "opacity_b" in leaves:
                op = op + leaves["opacity_b"].view(-1, 1)
            sp = torch.zeros_like(means, requires_grad=True) + 0
            if s.use_rgb:
                col, shs = leaves["shs"].squeeze(1), None
                if "color_w" in leaves:
                    w = leaves["color_w"].view(-1, 16, 3)
                    col = col * w[:, 0, :] + w[:, 1, :] - 1
                if "color_b" in leaves:
                    col = col + leaves["color_b"].view(-1, 16, 3)[:, 0, :]
            else:
                col, shs = None, leaves["shs"]
                if "color_w" in leaves:
                    shs = shs * leaves["color_w"].view(-1, 16, 3)
                if "color_b" in leaves:
                    shs = shs * leaves["color_w"].view(-1, 16, 3) + leaves["color_b"].view(-1, 16, 3)
            kw = dict(means3D=means, means2D=sp, opacities=op, scales=leaves["scaling"], rotations=leaves["rotation"], cov3D_precomp=None)
            img, _ = GaussianRasterizer(mk(s.bg, s.sh_degree))(shs=shs, colors_precomp=col, **kw)
            msk, _ = GaussianRasterizer(mk(zero, 0))(colors_precomp=torch.ones_like(means), **kw)
            loss = loss + (img - gt).abs().mean() + ((msk.mean(0) - gt[0]) ** 2).mean()
        loss.backward()

    import gc
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    gc.collect()          # (a full collection of the interpreter's ~1e6 objects is 40-60 ms: pending at the start of a 12-call window it
                          # was a third of the figure — measured with tools/ptr_probe2.py; collected here, never disabled)
    t0 = time.perf_counter()
    for _ in range(n_iter):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_iter * 1e3 / len(settings)


def self_launch(n: int) -> int:
    """Run this very command line as n ranks under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1)."""
    import socket
    import subprocess
    with socket.socket() as so:                       # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    env["GH_BENCH_LAUNCHER"] = "bench.py self-launch (torch.distributed.run, child processes)"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--views-per-step", type=int, default=8)
    ap.add_argument("--dist-backend", default=None, choices=[None, "nccl", "gloo"],
                    help="torch.distributed backend for N>1 (default: nccl = RCCL); gloo only to exercise the path without N GPUs")
    ap.add_argument("--config", default="two_hands")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="N>1: also all-reduce the per-Gaussian gradient block every step (what the sharded fit loop does); "
                         "default is the north star's protocol: independent views, RCCL for the scalar loss only")
    ap.add_argument("--pose-batch", action="store_true",
                    help="every view of a step is a DIFFERENT pose (its own Gaussian set, BASELINE configs[4] 'mixed poses'): "
                         "one launch sequence with GH_FLAG_PER_VIEW_GAUSSIANS instead of shared Gaussians seen by all views")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue the timed steps kernel by kernel instead of replaying one step (forward + loss + backward) "
                         "captured in a HIP graph; the eager loop is bound by the host for small batches or on a slow host")
    ap.add_argument("--graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank renders --views-per-step cameras; strong: the --views-per-step cameras of ONE "
                         "step are split over the ranks (BASELINE configs[3]: 8 novel views sharded over 8 GPUs = 1 view per GPU)")
    ap.add_argument("--split-streams", default="off", choices=["off", "on", "auto"],
                    help="GH_FLAG_SPLIT_STREAMS: render the step's views as two halves on two HIP streams inside the library "
                         "(bit-identical results); auto = from 4 views of more than half a megapixel per rank up")
    ap.add_argument("--defer-loss", default="on", choices=["on", "off"],
                    help="on (default): the fused loss's final sum runs as a spare workgroup of the render backward "
                         "(GH_FLAG_DEFER_LOSS_SUM) instead of a one-workgroup kernel between forward and backward; off: round 5's form")
    ap.add_argument("--fused-loss", default="on", choices=["on", "off"],
                    help="on (default): mean|render - gt| and its gradient come out of the render kernel's own epilogue (GhOutputs.l1_*); "
                         "off: gh_l1_loss reads the stored image back (rounds 1-4; same gradients bit for bit)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="EXPERIMENT: capture this many copies of the step on as many streams and replay them round-robin, so that "
                         "consecutive (independent) steps overlap on the GPU; 1 = steps strictly one after the other (the contract's default)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of --steps steps each; `value` / `ms_per_step` are the FIRST window's (the contract's exactly-K "
                         "steps), the others are reported as repeat statistics")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (the driver's own torchrun line, as CHILD
        # processes, before this process has made a single GPU call — never an exec of a process that touched the GPU), pass
        # rank 0's JSON line through and exit with the launcher's code. Replaces the reference's PL-DDP launch
        # (infer_one_shot.py:631, :638).
        sys.exit(self_launch(args.gpus))

    from guassianhand_amd import dist as ghdist
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.loss import rendered_l1_loss
    from guassianhand_amd.scenes import make_scene, perturbed_target_xyz
    import torch.distributed as tdist

    if args.dist_backend == "gloo":                     # functional check of the N>1 path on a box with fewer GPUs than ranks
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    rank, local, world = ghdist.init_from_env(args.dist_backend)
    if world != args.gpus:                              # fail loudly: a line that says n_gpus N must come from N ranks
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}; launch it as `python bench.py --gpus {args.gpus}` (self-"
              f"launching) or under torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a ROCm device (there is no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    rccl_ranks = None
    if world > 1:
        # one real collective before anything is measured: the line reports the number of ranks that ANSWERED it
        # (sum of ones over the process group), not the number that was asked for
        ones = torch.ones(1, device=torch.device("cuda", local))
        tdist.all_reduce(ones, op=tdist.ReduceOp.SUM)
        rccl_ranks = int(round(float(ones.item())))
        if rccl_ranks != args.gpus:
            print(f"bench.py: the all-reduce reached {rccl_ranks} ranks, --gpus says {args.gpus}", file=sys.stderr)
            sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    R.set_split_streams({"off": False, "on": True, "auto": "auto"}[args.split_streams])
    R.set_fused_loss(args.fused_loss == "on")
    V = args.views_per_step
    if args.scaling == "strong":
        assert V % world == 0, "--scaling strong: --views-per-step must be a multiple of the rank count"
        V_total, V = V, V // world                   # strong scaling: the step's cameras are dealt round-robin to the ranks
        scene_cpu = make_scene(args.config, n_views=V_total)
        mine = ghdist.shard_views(V_total, rank, world)
    else:
        V_total = V * world
        scene_cpu = make_scene(args.config, n_views=V_total)
        mine = [rank * V + i for i in range(V)]      # weak scaling: V views per rank
    scene_one = scene_cpu                            # one pose: what the CPU baseline renders
    if args.pose_batch:                              # V poses per rank: concatenate V different scenes, camera v of pose v
        import dataclasses
        from guassianhand_amd.scenes import SEED
        poses = [make_scene(args.config, n_views=V_total, seed=SEED + 1000 * (rank * V + b)) for b in range(V)]
        cat = lambda k: None if getattr(poses[0], k) is None else torch.cat([getattr(p_, k) for p_ in poses])
        scene_cpu = dataclasses.replace(poses[0], **{k: cat(k) for k in ("xyz", "opacity", "rotation", "scaling", "shs", "color_b", "opacity_b")})
    s = scene_cpu.to(dev)
    cams = s.cams()[mine].contiguous()
    H, W = s.H, s.W
    P = s.P // V if args.pose_batch else s.P
    pv = dict(per_view_gaussians=True) if args.pose_batch else {}

    # ground truth = render of a perturbed copy (positions + N(0, 1 mm)), forward only
    gt_xyz = perturbed_target_xyz(scene_cpu).to(dev)
    blend = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
    with torch.no_grad():
        gt, _ = R.rasterize_views(cams, gt_xyz, s.opacity, s.scaling, s.rotation, s.shs, H=H, W=W, use_rgb=s.use_rgb,
                                  sh_degree=s.sh_degree, sync=True, **blend, **pv)
    gt = gt.detach()

    names = ["xyz", "opacity", "scaling", "rotation", "shs"] + [k for k, v in blend.items() if v is not None]
    params = {"xyz": s.xyz, "opacity": s.opacity, "scaling": s.scaling, "rotation": s.rotation, "shs": s.shs}
    params.update({k: v for k, v in blend.items() if v is not None})
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}

    def local_step(sync: bool, defer_loss: bool = True):
        """One pass of the hot path over this rank's views: forward, loss, backward. No collective.
        defer_loss: the loss VALUE (logging / the collective; no gradient depends on it) is summed up by a spare workgroup of the
        render backward instead of a one-workgroup kernel between forward and backward (GH_FLAG_DEFER_LOSS_SUM): it exists
        once the step's backward has run, which is when every caller below looks at it."""
        for p in params.values():
            p.grad = None
        # render + loss as one autograd node: mean|img - gt| and dL/dimg from one fused pass (gh_l1_loss), dL/dloss applied
        # inside the render backward (GhGrads.upstream_scale) instead of in an elementwise pass over the images
        loss, _img, _ = rendered_l1_loss(cams, params["xyz"], params["opacity"], params["scaling"], params["rotation"],
                                         params["shs"], gt, H=H, W=W, use_rgb=s.use_rgb, sh_degree=s.sh_degree, sync=sync,
                                         xyz_b=params.get("xyz_b"), opacity_b=params.get("opacity_b"),
                                         color_w=params.get("color_w"), color_b=params.get("color_b"),
                                         defer_loss=defer_loss and args.defer_loss == "on", **pv)
        return loss

    def reduce_grads(loss):
        """Data-parallel fit (--allreduce-grads): sum the gradient block at the rasteriser boundary, ONE collective over the
        contiguous block the backward kernels wrote (rasterizer.last_grad_block): no packing pass, float 0 carries the loss."""
        block, n_red, _spans, _ = R.last_grad_block()
        work, buf = ghdist.allreduce_block(block, n_red, loss.detach())
        if work is not None:
            work.wait()
        return buf[0]

    def step(sync: bool):
        """Eager step: kernel-by-kernel enqueue. N>1, north star protocol (views are independent, RCCL only for the loss):
        the scalar all-reduce is issued as soon as the loss exists and runs on RCCL's own stream underneath the backward."""
        early = world > 1 and not args.allreduce_grads       # the loss collective is issued BEFORE the backward: it needs the value now
        loss = local_step(sync, defer_loss=not early)
        work = None
        if early:
            loss_sum = loss.detach().clone()
            work = tdist.all_reduce(loss_sum, op=tdist.ReduceOp.SUM, async_op=True)
        loss.backward(seed)
        if work is not None:
            work.wait()                              # stream-level dependency for RCCL (no host block); gloo blocks
            loss = loss_sum
        elif world > 1:
            loss = reduce_grads(loss)
        return loss

    seed = torch.ones((), dtype=torch.float32, device=dev)      # dL/dloss, allocated once (backward() would fill a new one per step)

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    # warm-up: the first step reads D back once to size the instance capacity, the rest are sync-free
    step(sync=True)
    for _ in range(max(0, args.warmup - 1)):
        step(sync=False)
    R.check_overflow()

    # The timed steps replay ONE captured step (forward + loss + backward of this rank's views) from a HIP graph: the
    # launch sequence is ~45 small kernels, and enqueueing them one by one leaves the result at the mercy of the host
    # (0.5 ms per step on a quiet box, several ms on a busy one). Collectives stay outside the graph.
    graph, g_loss, graph_note = None, None, None
    warm_replays = 0
    pipe = []
    if not args.no_graph:
        try:                                             # the leaves' AccumulateGrad nodes were created on the default stream
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass
        try:
            R.set_graph_mode(True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                local_step(sync=False).backward(seed)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread_local: HIP calls of other threads (e.g. the RCCL watchdog's event queries) must not invalidate the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                g_loss = local_step(sync=False)
                g_loss.backward(seed)
            # warm-up of the replay path itself (graph upload, clocks): untimed. At least --warmup replays, and at least
            # WARM_MS of them: the eager warm-up steps above are host-bound (the GPU idles between their kernels), and the first
            # ~30 ms of back-to-back replays run on a clock that is still ramping (measured: a window of 20 steps behind 5
            # replays read 0.662 ms per step where every later window read 0.640). `value` is the throughput of a loop that has
            # been running — what a training job sees —, and the line says how many replays preceded the timed region.
            t_w = time.perf_counter()
            warm_replays = 0
            while warm_replays < max(1, args.warmup) or (time.perf_counter() - t_w) * 1e3 < WARM_MS:
                graph.replay()
                warm_replays += 1
                if warm_replays % 8 == 0:
                    torch.cuda.synchronize()             # (the host would otherwise queue thousands of replays inside the time limit)
            torch.cuda.synchronize()
            pipe = []
            if args.pipeline > 1 and world == 1:
                for _ in range(args.pipeline):
                    st = torch.cuda.Stream()
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        local_step(sync=False).backward(seed)          # this stream's workspaces
                        gph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gph, stream=st, capture_error_mode="thread_local"):
                            gl = local_step(sync=False)
                            gl.backward(seed)
                    pipe.append((st, gph, gl))
                torch.cuda.synchronize()
        except Exception as e:                           # capture is an optimisation of the enqueue path, never a requirement
            graph, g_loss, graph_note = None, None, f"graph capture failed ({type(e).__name__}: {e}); eager steps"
            R.set_graph_mode(False)
            torch.cuda.synchronize()

    def replay_step(pending):
        """Graph step. N>1: the loss all-reduce of step k is issued after replay k and overlaps replay k+1."""
        graph.replay()
        if world == 1:
            return g_loss, None
        if args.allreduce_grads:
            return reduce_grads(g_loss), None
        if pending is not None:
            pending.wait()
        loss_sum = g_loss.detach().clone()
        return loss_sum, tdist.all_reduce(loss_sum, op=tdist.ReduceOp.SUM, async_op=True)

    def timed_window():
        barrier()
        t0 = time.perf_counter()
        pending, loss = None, None
        for k in range(args.steps):
            if graph is not None and args.pipeline > 1 and world == 1:
                st, gph, gl = pipe[k % len(pipe)]
                with torch.cuda.stream(st):
                    gph.replay()
                loss = gl
            elif graph is not None:
                loss, pending = replay_step(pending)
            else:
                loss = step(sync=False)
        if pending is not None:
            pending.wait()
        t_enq = time.perf_counter() - t0             # host time to enqueue the timed steps (GPU-bound if << dt)
        barrier()
        return time.perf_counter() - t0, t_enq, loss

    dt, t_enq, loss = timed_window()                 # THE timed region: exactly --steps steps between two barriers
    extra = [timed_window()[0] for _ in range(max(0, args.repeats - 1))]      # repeat statistics (not `value`)
    R.check_overflow()
    # per-step spread (>= 20 timed iterations, median and p10 / p90): every step bracketed by its own HIP events
    step_ms = []
    if world == 1:
        evs = []
        for _ in range(max(20, args.steps)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if graph is not None:
                graph.replay()
            else:
                step(sync=False)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        step_ms = sorted(a.elapsed_time(b) for a, b in evs)
        R.check_overflow()
    if graph is not None:
        R.set_graph_mode(False)

    # Stage times for the roofline leg: HIP events around the stage entry points (same stream) on eager steps of the same
    # workload, run right after the timed region (events cannot sit inside the captured graph, and the timed steps stay
    # free of the extra event records).
    stage_ms = {}
    if not args.no_stage_timing:
        R.enable_stage_timing(True)
        for _ in range(args.steps):
            local_step(sync=False).backward(seed)
        stage_ms = R.stage_timing_summary()
        R.enable_stage_timing(False)
        R.check_overflow()

    if world > 1:
        tmax = torch.tensor([dt] + extra, dtype=torch.float64, device=dev)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        dt, extra = float(tmax[0].item()), [float(x) for x in tmax[1:].tolist()]
    renders = args.steps * V * world
    value = renders / dt
    # checksum of the step's result (after the last timed step): the N-rank control tests compare it with a 1-rank run
    grad_l1 = float(sum(params[k].grad.detach().abs().sum() for k in names if params[k].grad is not None))

    strong_info = None
    if args.scaling == "strong":
        # predicted step = the committed single-GPU step of the SAME per-rank view count (profiles/<tag>_bench_<V>view.json);
        # the difference to `ms_per_step` is what the collective and the launch skew cost
        pred = None
        f = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_bench_{V}view.json" if V != 8 else f"{PROFILE_TAG}_bench_default.json")
        if args.config == "two_hands" and not args.pose_batch and os.path.exists(f):
            try:
                pred = json.load(open(f))["ms_per_step"]
            except (KeyError, ValueError):
                pred = None
        strong_info = {"views_total": V_total, "views_per_rank": V, "predicted_ms_per_step": pred,
                       "predicted_from": os.path.relpath(f, ROOT) if pred is not None else None,
                       "measured_ms_per_step": dt / args.steps * 1e3}
    if rank == 0:
        # instance count of the published algorithm (every tile of the 3-sigma rects) beside the exactly culled one
        with torch.no_grad():
            col = dict(colors_precomp=s.shs.reshape(-1, 3)) if s.use_rgb else dict(shs=s.shs, sh_degree=s.sh_degree)
            _, _, rctx = R.raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, sync=True, split_streams=False,
                                           **col, **blend, **pv)
            rect = R.workspace_views(rctx)["rect"].long()
            D_rect = int((((rect >> 16 & 255) - (rect & 255)) * ((rect >> 24 & 255) - (rect >> 8 & 255))).sum())
            del rctx
        D = R.last_num_rendered()
        ab = algorithmic_bytes(P, V, H, W, D, C=12 if s.use_rgb else 192)
        roofline = None
        stages = {}
        if stage_ms:
            for k, ms in stage_ms.items():
                stages[k] = {"ms": ms, "alg_GBs": ab[k] / (ms * 1e-3) / 1e9 if ms > 0 else None}
            single = {k: v for k, v in stage_ms.items() if k != "binning"}   # binning is a multi-kernel stage
            dom = max(single, key=single.get)
            ach = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9
            kname = "gh_" + dom + "_kernel"
            tr, tr_src = pmc_profile("traffic", kname, args, V)
            sq, sq_src = pmc_profile("sq", kname, args, V)
            roofline = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr["traffic_bytes"] if tr else None,
                        "traffic_source": tr_src,
                        "alg_bytes_per_launch": ab[dom], "ms_per_launch": stage_ms[dom],
                        "timing": f"HIP events around the stage entry points, {args.steps} eager steps run right after the timed steps",
                        # what actually bounds the kernel (it is not bandwidth): vector-ALU issue and the LDS pipe, from the
                        # committed SQ counter passes — arithmetic in profiles/r4_pmc_sq*.json
                        "secondary": sq["secondary"] if sq else None, "secondary_source": sq_src}
        out = {
            "metric": "fwd+bwd renders/sec @512x334, ~100k Gaussians", "value": value, "unit": "renders/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: P={P} Gaussians, {H}x{W}, "
                                   f"{'RGB colours' if s.use_rgb else 'SH degree %d colours' % s.sh_degree}, attribute blend "
                                   f"{'on' if s.color_w is not None else 'off'}"
                                   + (" (BASELINE configs[2])" if args.config == "two_hands" and not args.pose_batch else "")
                                   + (", pose batch: every view its own Gaussian set" if args.pose_batch else ""),
                       "views_per_step_per_gpu": V, "instances_per_step_per_gpu": D,
                       "instances_in_3sigma_rects": D_rect, "parallelism": f"view-parallel x{world}",
                       "collective": None if world == 1 else ("all-reduce(loss + gradient block)" if args.allreduce_grads
                                                              else "all-reduce(loss)"),
                       "loss": "mean|img-gt|", "final_loss": float(loss.detach()), "grad_l1": grad_l1,
                       "views_per_step_total": V * world,
                       "ranks": {"world_size": world, "answered_all_reduce": rccl_ranks,
                                 "backend": None if world == 1 else tdist.get_backend(),
                                 "launched_by": os.environ.get("GH_BENCH_LAUNCHER", "external launcher" if world > 1 else "single process")},
                       "strong_scaling": strong_info,
                       "repeats": {"windows": 1 + len(extra), "steps_per_window": args.steps,
                                   "ms_per_step_min": min([dt] + extra) / args.steps * 1e3,
                                   "ms_per_step_median": sorted([dt] + extra)[len([dt] + extra) // 2] / args.steps * 1e3,
                                   "ms_per_step_max": max([dt] + extra) / args.steps * 1e3,
                                   "note": "`value` / `ms_per_step` are window 1 (the contract's timed region); the others follow it"},
                       "step_ms": None if not step_ms else {
                           "n": len(step_ms), "median": step_ms[len(step_ms) // 2], "p10": step_ms[len(step_ms) // 10],
                           "p90": step_ms[(len(step_ms) * 9) // 10], "how": "HIP events around every single step, after the timed windows"},
                       "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
                       "split_streams": bool(V >= 2 and (R._split_policy is True or (
                           R._split_policy == "auto" and V >= 4 and H * W > R._SPLIT_AUTO_MIN_PIXELS))),
                       "pipelined_steps": args.pipeline, "fused_loss": args.fused_loss == "on", "deferred_loss_sum": args.defer_loss == "on",
                       "warmup_done": {"eager_steps": max(1, args.warmup), "graph_replays": warm_replays,
                                       "note": f"--warmup is a minimum: replays continue until {WARM_MS:.0f} ms have passed, so that the timed "
                                               "region does not start on a ramping clock"},
                       "hip_graph": graph is not None, "timed_steps": "HIP graph replay of one captured step" if graph is not None
                       else "eager kernel-by-kernel enqueue" + (f" [{graph_note}]" if graph_note else "")},
            "roofline": roofline, "stages": stages,
        }
        if world == 1 and not args.no_cpu_baseline:
            if not args.pose_batch:
                # the reference's own 2-call protocol through the drop-in, per view, beside the fused per-view cost —
                # a HOST-bound figure, taken before the CPU baseline has 16 OpenMP threads spinning beside this one
                out["config"]["two_call_ms_per_view"] = two_call_cost(s, list(mine), n_iter=max(8, 30 // len(mine)))   # this step's views, one backward
                out["config"]["two_call_ms_per_view_one_view_steps"] = two_call_cost(s, [mine[0]])      # B = 1, one view per step, a backward each
                out["config"]["fused_ms_per_view"] = dt / args.steps * 1e3 / V
            out["cpu_baseline"] = cpu_baseline(scene_one, args.cpu_seconds, torch_reference=(args.config == "random1k"))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
