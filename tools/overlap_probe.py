"""Does splitting the 8-view step into two 4-view halves on two HIP streams (one captured graph, fork/join) hide the
latency-bound binning launches of one half under the render kernels of the other?  Prints ms per 8-view step for
1 stream x 8 views, 2 streams x 4 views, 4 streams x 2 views."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.loss import rendered_l1_loss
from guassianhand_amd.scenes import make_scene, perturbed_target_xyz

dev = torch.device("cuda:0")
sc_cpu = make_scene("two_hands", n_views=8)
s = sc_cpu.to(dev)
cams = s.cams().contiguous()
H, W = s.H, s.W
gt_xyz = perturbed_target_xyz(sc_cpu).to(dev)
blend = dict(xyz_b=s.xyz_b, opacity_b=s.opacity_b, color_w=s.color_w, color_b=s.color_b)
with torch.no_grad():
    gt, _ = R.rasterize_views(cams, gt_xyz, s.opacity, s.scaling, s.rotation, s.shs, H=H, W=W, use_rgb=s.use_rgb,
                              sh_degree=s.sh_degree, sync=True, **blend)
gt = gt.detach()
seed = torch.ones((), device=dev)
try:
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
except AttributeError:
    pass


def mk_params():
    p = {"xyz": s.xyz, "opacity": s.opacity, "scaling": s.scaling, "rotation": s.rotation, "shs": s.shs}
    p.update(blend)
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


def half_step(params, views):
    for p in params.values():
        p.grad = None
    loss, _, _ = rendered_l1_loss(views[0], params["xyz"], params["opacity"], params["scaling"], params["rotation"],
                                  params["shs"], views[1], H=H, W=W, use_rgb=s.use_rgb, sh_degree=s.sh_degree,
                                  sync=False, xyz_b=params["xyz_b"], opacity_b=params["opacity_b"], color_w=params["color_w"],
                                  color_b=params["color_b"])
    loss.backward(seed)
    return loss


def fwd_only(params, views):
    for p in params.values():
        p.grad = None
    loss, _, _ = rendered_l1_loss(views[0], params["xyz"], params["opacity"], params["scaling"], params["rotation"],
                                  params["shs"], views[1], H=H, W=W, use_rgb=s.use_rgb, sh_degree=s.sh_degree,
                                  sync=False, xyz_b=params["xyz_b"], opacity_b=params["opacity_b"], color_w=params["color_w"],
                                  color_b=params["color_b"])
    return loss


def run(nsplit, stagger=False, midjoin=False):
    per = 8 // nsplit
    groups = [(cams[i * per:(i + 1) * per].contiguous(), gt[i * per:(i + 1) * per].contiguous()) for i in range(nsplit)]
    plist = [mk_params() for _ in groups]
    streams = [torch.cuda.Stream() for _ in groups[1:]]
    # warm-up (sizes the capacities)
    for g, p in zip(groups, plist):
        for _ in range(3):
            half_step(p, g)
    torch.cuda.synchronize()
    R.check_overflow()
    R.set_graph_mode(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for g, p in zip(groups, plist):
            half_step(p, g)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        if midjoin:                                   # both halves join after the forward + loss, fork again for the backward
            streams[0].wait_stream(cur)
            with torch.cuda.stream(streams[0]):
                lossB = fwd_only(plist[1], groups[1])
            lossA = fwd_only(plist[0], groups[0])
            cur.wait_stream(streams[0])
            streams[0].wait_stream(cur)
            with torch.cuda.stream(streams[0]):
                lossB.backward(seed)
            lossA.backward(seed)
            cur.wait_stream(streams[0])
        elif stagger:                                   # B starts when A's forward is done: B's binning under A's backward
            lossA = fwd_only(plist[0], groups[0])
            streams[0].wait_stream(cur)
            with torch.cuda.stream(streams[0]):
                half_step(plist[1], groups[1])
            lossA.backward(seed)
            cur.wait_stream(streams[0])
        else:
            for st in streams:
                st.wait_stream(cur)
            for st, g, p in zip(streams, groups[1:], plist[1:]):
                with torch.cuda.stream(st):
                    half_step(p, g)
            half_step(plist[0], groups[0])
            for st in streams:
                cur.wait_stream(st)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    R.set_graph_mode(False)
    R.check_overflow()
    print(f"{nsplit} stream(s) x {per} views{' (staggered)' if stagger else ''}{' (join after forward)' if midjoin else ''}: {best:.3f} ms per 8-view step", flush=True)


run(1); run(2); run(2, midjoin=True); run(2); run(2, midjoin=True); run(1)
