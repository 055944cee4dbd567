"""The forward's fine-grained form (gh_render_fwd_kernel<.., FINE>, gh_fwd_consume_fine: wave = 2x2 pixels x 16 depth slots) against the
oracle, with EVERY tile forced into it — by default only the 32 heaviest tiles of a small launch with lists of 256+ entries go fine,
which the suite's small scenes rarely have. The library reads its two thresholds from the environment once per process
(GH_FWD_FINE_K / GH_FWD_FINE_MIN), so each mode runs in a child process (one at a time: the box allows few processes on the card).

What must hold: images, n_contrib, final T bit-equal to the oracle in either form (the same arithmetic per pixel in the same order);
gradients within 1e-3; the fused image loss bit-IDENTICAL whichever tiles went fine (which tiles do is a scheduling decision that
depends on the previous call over the workspace — the loss may not depend on it). The same for the other scheduling hints the
workspace carries between calls (GH_FWD_HEAVY_ORDER: the launch order; GH_BWD_CLASSES: the backward's work list by measured cost):
loss, image and gradients of four consecutive steps are the same bits in every mode."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, hashlib
import torch
sys.path.insert(0, %(root)r)
from guassianhand_amd.scenes import make_scene, ring_cameras
from guassianhand_amd.loss import rendered_l1_loss
from tests.test_gpu_parity import compare
dev = torch.device("cuda:0")
out = {}
def h(t): return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
# (1) the whole parity comparison (every stage array, image bit-equal to the oracle, n_contrib, final T, gradients <= 1e-3)
def sized(sc, nv, H, W, f):
    sc.H, sc.W = H, W
    sc.w2c, sc.K = ring_cameras(sc.xyz.mean(0), nv, H, W, f)
    return sc
cases = [make_scene("random1k", n_views=2, P=3000), make_scene("random1k", n_views=1, P=6000, use_rgb=False, blend=True),
         sized(make_scene("random1k", n_views=3, P=2500, use_rgb=True, blend=True), 3, 77, 131, 1.3 * 131),
         sized(make_scene("two_hands", n_views=2), 2, 256, 256, 650.0), make_scene("one_hand", n_views=1, P=20000, scale_mean=-5.0)]
for ci, sc in enumerate(cases):
    out["D%%d" %% ci] = compare(sc, dev)
# (2) the fused image loss: bit-identical whatever form walked the tiles (compared across the child processes by the parent)
for ci, sc in enumerate([sized(make_scene("two_hands", n_views=2), 2, 256, 256, 650.0), sized(make_scene("random1k", n_views=1, P=6000), 1, 100, 90, 130.0)]):
    s = sc.to(dev)
    cams = sc.cams().to(dev)
    gt = torch.rand(cams.shape[0], 3, sc.H, sc.W, generator=torch.Generator().manual_seed(3)).to(dev)
    xyz = s.xyz.clone().requires_grad_(True)
    kw = dict(H=sc.H, W=sc.W, use_rgb=sc.use_rgb, sh_degree=sc.sh_degree)
    for rep in range(3):          # (the launch order and the choice of fine tiles follow the previous call: three calls, one loss)
        loss, img, _ = rendered_l1_loss(cams, xyz, s.opacity, s.scaling, s.rotation, s.shs, gt, **kw)
        out["loss%%d_%%d" %% (ci, rep)] = h(loss)
        out["img%%d_%%d" %% (ci, rep)] = h(img)
    loss.backward()
    out["dxyz%%d" %% ci] = h(xyz.grad)
# (3) a launch in the range of the backward's cost classes (2,049 .. 8,192 tiles: four views of 512x334): forward + backward four
#     times over one workspace — the second step on, the launch order comes from the first step's measurements and the backward's work
#     list from what its items cost the step before. Loss, image and gradient must be the same bits in every step and every mode.
sc = make_scene("two_hands", n_views=4)
s = sc.to(dev)
cams = sc.cams().to(dev)
gt = torch.rand(4, 3, sc.H, sc.W, generator=torch.Generator().manual_seed(4)).to(dev)
xyz = s.xyz.clone().requires_grad_(True)
kw = dict(H=sc.H, W=sc.W, use_rgb=sc.use_rgb, sh_degree=sc.sh_degree)
for rep in range(4):
    xyz.grad = None
    loss, img, _ = rendered_l1_loss(cams, xyz, s.opacity, s.scaling, s.rotation, s.shs, gt, **kw)
    loss.backward()
    out["loss4_%%d" %% rep] = h(loss); out["img4_%%d" %% rep] = h(img); out["dxyz4_%%d" %% rep] = h(xyz.grad)
print("RESULT " + json.dumps(out))
'''


def _run(env_over):
    env = dict(os.environ)
    env.update(env_over)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_every_tile_in_the_fine_form_equals_the_oracle_and_the_loss_does_not_depend_on_the_form():
    fine = _run({"GH_FWD_FINE_K": "1000000", "GH_FWD_FINE_MIN": "0"})        # every tile of every small launch goes fine
    coarse = _run({"GH_FWD_FINE_K": "0"})                                      # none does
    mixed = _run({"GH_FWD_FINE_K": "7", "GH_FWD_FINE_MIN": "64"})            # a few do — which ones depends on the previous call
    plain = _run({"GH_FWD_HEAVY_ORDER": "0", "GH_BWD_CLASSES": "0"})          # launch order by list length, the work list in one piece
    for k in fine:
        assert fine[k] == plain[k], (k, fine[k], plain[k])                    # no scheduling hint changes a bit of any result
    for k in fine:
        if k.startswith(("loss", "img", "dxyz4")):
            base = k.rsplit("_", 1)[0]
            assert fine[k] == fine[base + "_0"] == coarse[k] == mixed[k], (k, fine[k], coarse[k], mixed[k])
        elif not k.startswith("dxyz"):            # (instance counts; the gradient sums are compared below)
            assert fine[k] == coarse[k] == mixed[k], (k, fine[k], coarse[k], mixed[k])
    # the backward is the same kernel over the same forward state in every form: its gradients are the same bits
    for k in (k for k in fine if k.startswith("dxyz") and not k.startswith("dxyz4")):
        assert fine[k] == coarse[k] == mixed[k], (k, fine[k], coarse[k], mixed[k])


def test_the_launch_order_hint_is_only_used_for_the_cameras_that_left_it():
    """rasterizer._forward_full sets GH_FLAG_FRESH_ORDER unless the workspace it got was last used with the SAME camera tensor (object
    and version): a recycled address or an in-place update of the cameras does not count as the same. The image is the same bits
    either way (the flag moves the tile ranking from the projection kernel's spare workgroups back into a kernel of its own)."""
    import torch
    from guassianhand_amd import _abi
    from guassianhand_amd.rasterizer import raster_forward, clear_workspace_pool
    from guassianhand_amd.scenes import make_scene
    from tests.helpers import scene_kwargs
    dev = torch.device("cuda:0")
    sc = make_scene("random1k", n_views=2, P=3000)
    s = sc.to(dev)
    kw, bl = scene_kwargs(s)
    cams = sc.cams().to(dev)
    clear_workspace_pool()

    def render(c):
        img, _, ctx = raster_forward(c, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=True, **kw, **bl)
        fresh = bool(ctx.dims.flags & _abi.GH_FLAG_FRESH_ORDER)
        del ctx                                           # (the workspace goes back to the pool: the next call gets this buffer)
        return img, fresh

    img0, f0 = render(cams)
    img1, f1 = render(cams)
    assert f0 and not f1                                  # a buffer fresh from the allocator; then the same cameras again
    cams2 = cams.clone()
    img2, f2 = render(cams2)
    img3, f3 = render(cams2)
    assert f2 and not f3                                  # other tensor object (same values): stale; then valid again
    cams2.mul_(1.0)                                       # in-place update: the version counter moves
    img4, f4 = render(cams2)
    assert f4
    for im in (img1, img2, img3, img4):
        assert torch.equal(im, img0)
