"""Two host threads, two streams, one device, the drop-in's two-call protocol with backward: the rasteriser's
per-device state (capacities, pending read-backs, workspace pool, the geometry record of the RGB -> mask pair) is touched from
both threads and from autograd's backward thread. Every image and gradient must equal the single-threaded result bit for bit."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _protocol(sc, dev, cam):
    """forward_single_view's two rasteriser calls + a backward; returns (rgb, mask, grads)."""
    from tests.helpers import forward_single_view
    from guassianhand_amd.renderer import GaussianModel
    leaves = [t.to(dev).clone().requires_grad_(True) for t in (sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)]
    gs = GaussianModel(*leaves)
    r = forward_single_view(gs, cam, torch.zeros(3, device=dev), use_rgb=True)
    (r["comp_rgb"].sum() * 0.5 + (r["comp_mask"] ** 2).sum()).backward()
    return r["comp_rgb"].detach().clone(), r["comp_mask"].detach().clone(), [t.grad.clone() for t in leaves]


def test_two_threads_two_streams_through_the_drop_in(dev):
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.camera import Camera
    from guassianhand_amd.scenes import make_scene
    scenes = [make_scene("random1k", n_views=1, P=900, seed=3), make_scene("random1k", n_views=1, P=1400, seed=4)]
    scenes[1].H, scenes[1].W = 96, 80
    cams = [Camera.from_w2c(sc.w2c[0].to(dev), sc.K[0].to(dev), sc.H, sc.W) for sc in scenes]
    want = [_protocol(sc, dev, cam) for sc, cam in zip(scenes, cams)]
    torch.cuda.synchronize()
    errors, n_iter = [], 12

    def worker(k):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for it in range(n_iter):
                    rgb, msk, grads = _protocol(scenes[k], dev, cams[k])
                    stream.synchronize()
                    if not (torch.equal(rgb, want[k][0]) and torch.equal(msk, want[k][1])):
                        errors.append(f"thread {k} iteration {it}: image differs")
                    for a, b in zip(grads, want[k][2]):
                        if not torch.equal(a, b):
                            errors.append(f"thread {k} iteration {it}: gradient differs")
        except Exception as e:                       # noqa: BLE001 — reported below, with the thread it came from
            errors.append(f"thread {k}: {type(e).__name__}: {e}")

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    R.check_overflow()
    assert not errors, errors[:5]
