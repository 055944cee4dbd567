"""One-shot fit loop host logic on CPU: the rasteriser is replaced by the dense autograd oracle
(tests/helpers.oracle_render_views, same signature as render_views), so this covers the map lookups, the loss of
utils.py:180-291 / infer_one_shot.py:514-519, Adam + MultiStepLR, and — with gloo, world size 2 — the camera
sharding with the single all-reduce at the rasteriser boundary."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from guassianhand_amd import fit as F
from tests.helpers import oracle_render_views, tiny_fit_problem


def make_fit(pb):
    f = F.OneShotFit(pb["gs"], pb["uv"], use_rgb=True, map_hw=pb["map_hw"], lr=0.01, render_fn=oracle_render_views)
    return f


def targets(pb):
    f = make_fit(pb)
    with torch.no_grad():
        f.color_w.copy_(pb["true"]["color_w"]); f.color_b.copy_(pb["true"]["color_b"]); f.opacity_b.copy_(pb["true"]["opacity_b"])
        out = f.render(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], f.blend_values())
    return out["comp_rgb"].detach(), out["comp_mask"].detach().mean(-1)


def test_sample_map_matches_reference_grid_sample_convention():
    tex = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    uv = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [1.0, -1.0]])
    out = F.sample_map(tex, uv)
    assert out.shape == (4, 2)
    assert torch.equal(out[0], tex[:, 0, 0]) and torch.equal(out[1], tex[:, 2, 4])     # align_corners=True corners
    assert torch.allclose(out[2], tex[:, 1, 2]) and torch.equal(out[3], tex[:, 0, 4])  # uv[...,0] is x (width)


def test_loss_terms_match_reference_formulas():
    g = torch.Generator().manual_seed(1)
    rgb, gt = torch.rand(2, 8, 8, 3, generator=g), torch.rand(2, 8, 8, 3, generator=g)
    mask3 = torch.rand(2, 8, 8, 1, generator=g).expand(-1, -1, -1, 3) * 1.2
    gm = (torch.rand(2, 8, 8, generator=g) > 0.5).float()
    want = sum(10.0 * (rgb[v] - gt[v]).abs().mean() +
               1.0 * torch.nn.functional.mse_loss(mask3[v].mean(-1).clip(-0.001, 1.0), gm[v]) for v in range(2))
    assert torch.allclose(F.fit_loss(rgb, mask3, gt, gm), want, rtol=1e-6)


def test_fit_reduces_loss_and_schedule():
    pb = tiny_fit_problem()
    gt_rgb, gt_mask = targets(pb)
    f = make_fit(pb)
    losses = [float(f.step(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)) for _ in range(25)]
    assert losses[-1] < 0.7 * losses[0], losses
    assert float((f.color_w - 1).abs().max()) > 0 and float(f.color_b.abs().max()) > 0 and float(f.opacity_b.abs().max()) > 0
    assert f.color_b.shape == (48, *pb["map_hw"]) and f.opacity_b.shape == (1, *pb["map_hw"])        # reference layout views
    assert f.xyz_b.grad is None and float(f.xyz_b.abs().max()) == 0.0                  # frozen like the reference
    lrs = []
    for _ in range(6):
        f.end_epoch(); lrs.append(f.opt.param_groups[0]["lr"])
    assert lrs == pytest.approx([0.01, 0.005, 0.005, 0.005, 0.0025, 0.0025])          # milestones 2, 5 (gamma 0.5)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from guassianhand_amd import dist as ghdist
    ghdist.init_from_env(backend="gloo")
    torch.set_num_threads(2)
    pb = tiny_fit_problem()
    gt_rgb, gt_mask = targets(pb)
    f = make_fit(pb)
    losses = [float(f.step(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)) for _ in range(3)]
    if rank == 0:
        q.put((losses, f.color_w.detach().numpy().copy(), f.color_b.detach().numpy().copy(), f.opacity_b.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_fit_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses2, cw2, cb2, ob2 = q.get(timeout=500)
    cw2, cb2, ob2 = (torch.from_numpy(a) for a in (cw2, cb2, ob2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pb = tiny_fit_problem()
    gt_rgb, gt_mask = targets(pb)
    f = make_fit(pb)
    losses1 = [float(f.step(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)) for _ in range(3)]
    assert losses2 == pytest.approx(losses1, rel=1e-4)
    assert torch.allclose(cw2, f.color_w.detach(), atol=2e-4)
    assert torch.allclose(cb2, f.color_b.detach(), atol=2e-4) and torch.allclose(ob2, f.opacity_b.detach(), atol=2e-4)
