"""View-parallel sharding on CPU with gloo, world_size 2: cameras are dealt round-robin,
the ONE collective is an all-reduce(sum) of the fused per-Gaussian gradient block + the scalar loss, and the
reduced result equals the single-process result over all views. The per-view 'render' here is the CPU
oracle (checker) — the HIP path needs a GPU — so this test covers the sharding/collective logic only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from guassianhand_amd import dist as ghdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _per_view_grads(view):
    """Deterministic stand-in for the gradients one rank computes for one camera."""
    from guassianhand_amd.scenes import make_scene
    from oracle.oracle_c import OracleRender
    sc = make_scene("random1k", n_views=4, P=150)
    cams = sc.cams()[view:view + 1]
    o = OracleRender(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, colors_precomp=sc.shs.squeeze(1))
    g = torch.Generator().manual_seed(100 + view)
    dimg = torch.randn(1, 3, sc.H, sc.W, generator=g)
    grads = o.backward(dimg)
    grads.pop("means2D")
    loss = (o.image * dimg).sum()
    return loss, grads


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = ghdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    views = ghdist.shard_views(4, rank, world)
    loss, grads = None, None
    for v in views:
        l, g = _per_view_grads(v)
        loss = l if loss is None else loss + l
        grads = g if grads is None else {k: grads[k] + g[k] for k in g}
    order = sorted(grads)
    # (a) the packing form (CPU / gloo tensors), (b) the in-place form over ONE contiguous block laid out like
    # rasterizer.last_grad_block(): [4 caller floats | parts at 16-byte boundaries], float 0 = loss
    off, spans = 4, []
    for k in order:
        spans.append((k, grads[k].shape, off, grads[k].numel()))
        off += (grads[k].numel() + 3) & ~3
    block = torch.zeros(off)
    for k, shp, a, n in spans:
        block[a:a + n] = grads[k].reshape(-1)
    work, buf = ghdist.allreduce_block(block, off, loss)
    if work is not None:
        work.wait()
    loss_b, grads_b = buf[0].clone(), {k: buf[a:a + n].view(shp).clone() for k, shp, a, n in spans}
    loss, grads = ghdist.allreduce_grads(grads, loss, order)
    assert torch.equal(loss_b, loss.reshape(()).float()) and all(torch.equal(grads_b[k], grads[k]) for k in order)
    if rank == 0:
        q.put((float(loss), {k: v.numpy().copy() for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_views_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(v for r in range(world) for v in ghdist.shard_views(8, r, world))
        assert seen == list(range(8))
        assert all(len(ghdist.shard_views(8, r, world)) == 8 // world for r in range(world))


def test_pack_unpack_roundtrip():
    g = {"a": torch.arange(6.0).reshape(2, 3), "b": torch.ones(4)}
    buf, meta = ghdist.pack_grads(g, torch.tensor(2.5), ["a", "b"])
    assert buf.numel() == 1 + 6 + 4
    loss, out = ghdist.unpack_grads(buf, meta)
    assert float(loss) == 2.5 and torch.equal(out["a"], g["a"]) and torch.equal(out["b"], g["b"])


@pytest.mark.timeout(300)
def test_two_rank_allreduce_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss2, grads2 = q.get(timeout=240)
    grads2 = {k: torch.from_numpy(v) for k, v in grads2.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over all 4 views
    loss1, grads1 = None, None
    for v in range(4):
        l, g = _per_view_grads(v)
        loss1 = l if loss1 is None else loss1 + l
        grads1 = g if grads1 is None else {k: grads1[k] + g[k] for k in g}
    assert loss2 == pytest.approx(float(loss1), rel=1e-5)
    for k in grads1:
        assert torch.allclose(grads2[k], grads1[k], rtol=1e-5, atol=1e-6), k
