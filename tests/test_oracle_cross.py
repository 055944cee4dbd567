"""Oracle A (dense PyTorch, autograd-derived backward) vs Oracle B (C, hand-written backward) and float64
finite differences of Oracle A (SURVEY.md §8c viii-ix). This is what validates the backward FORMULAS
independently of any hand-written chain rule."""
import pytest
import torch

from guassianhand_amd.scenes import make_scene
from oracle import oracle_torch as OT
from oracle.oracle_c import OracleRender
from tests.helpers import dimg_like, rel_l2


def dense(sc, view, tensors, dtype=torch.float32, blend=None):
    c = sc.cams()[view].to(dtype)
    xyz, op, sca, rot, shs = tensors
    means, opac, cols, sh = OT.blend_attributes(xyz, op, shs, use_rgb=sc.use_rgb, **(blend or {}))
    kw = dict(colors_precomp=cols) if sc.use_rgb else dict(shs=sh, sh_degree=sc.sh_degree)
    return OT.rasterize_dense(means, opac, sca, rot, viewmatrix=c[:16].reshape(4, 4), projmatrix=c[16:32].reshape(4, 4),
                              campos=c[32:35], tanfovx=float(c[35]), tanfovy=float(c[36]), bg=c[37:40], H=sc.H, W=sc.W, **kw)


@pytest.mark.parametrize("use_rgb,blend", [(True, False), (False, False), (True, True), (False, True)])
def test_oracle_a_vs_b_image_and_grads(use_rgb, blend):
    sc = make_scene("random1k", n_views=2, P=400, use_rgb=use_rgb, blend=blend)
    if blend:
        g = torch.Generator().manual_seed(3)
        sc.xyz_b = 0.003 * torch.randn(3, generator=g)
    names = ["xyz", "opacity", "scaling", "rotation", "shs"]
    leaves = [getattr(sc, n).clone().requires_grad_(True) for n in names]
    bl = None
    if blend:
        bl = {k: getattr(sc, k).clone().requires_grad_(True) for k in ("color_w", "xyz_b", "color_b", "opacity_b")}
    dimg = dimg_like(2, sc.H, sc.W)
    imgs = []
    loss = 0
    for v in range(2):
        img, radii = dense(sc, v, leaves, blend=bl)
        imgs.append(img)
        loss = loss + (img * dimg[v]).sum()
    loss.backward()
    kw = dict(colors_precomp=sc.shs.squeeze(1)) if use_rgb else dict(shs=sc.shs, sh_degree=3)
    blc = {k: getattr(sc, k) for k in ("xyz_b", "opacity_b", "color_w", "color_b")} if blend else {}
    orc = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, **kw, **blc)
    assert (torch.stack(imgs) - orc.image).abs().max() < 5e-6
    gb = orc.backward(dimg)
    pairs = [("means3D", leaves[0]), ("opacities", leaves[1]), ("scales", leaves[2]), ("rotations", leaves[3]),
             ("colors_precomp" if use_rgb else "shs", leaves[4])]
    if blend:
        pairs += [(k, bl[k]) for k in ("xyz_b", "opacity_b", "color_w", "color_b")]
    for k, leaf in pairs:
        assert rel_l2(gb[k], leaf.grad) < 2e-5, k


def test_oracle_a_finite_differences_float64():
    """Central differences of the float64 dense oracle on a scene without borderline decisions."""
    sc = make_scene("random1k", n_views=1, P=60)
    d = torch.float64
    base = [getattr(sc, n).to(d) for n in ("xyz", "opacity", "scaling", "rotation", "shs")]
    base[2] = base[2] * 3.0                      # fatter Gaussians -> more overlap
    dimg = dimg_like(1, sc.H, sc.W, seed=9)[0].to(d)

    def f(ts):
        img, _ = dense(sc, 0, ts, dtype=d)
        return (img * dimg).sum()

    leaves = [t.clone().requires_grad_(True) for t in base]
    f(leaves).backward()
    g = torch.Generator().manual_seed(2)
    for ti, name in enumerate(("xyz", "opacity", "scaling", "rotation", "shs")):
        flat = base[ti].reshape(-1)
        idx = torch.randint(0, flat.numel(), (6,), generator=g)
        for j in idx.tolist():
            h = 1e-6 * max(1.0, abs(float(flat[j])))
            if name == "scaling":
                h = 1e-7
            vals = []
            for sgn in (+1, -1):
                ts = [t.clone() for t in base]
                ts[ti].reshape(-1)[j] += sgn * h
                vals.append(float(f(ts)))
            fd = (vals[0] - vals[1]) / (2 * h)
            an = float(leaves[ti].grad.reshape(-1)[j])
            assert fd == pytest.approx(an, rel=2e-4, abs=1e-6), (name, j)


@pytest.mark.parametrize("use_rgb,blend", [(True, True), (False, True), (True, False)])
def test_baseline_mode_equals_the_checker(use_rgb, blend):
    """gho_set_parallel(1) (bench.py's cpu_baseline leg: emit / per-tile sort / chain rule under OpenMP) against the serial
    checker path: identical lists, image, state; per-Gaussian gradients bit for bit (the views of a Gaussian are summed in the
    same order); the global (48,) color_w gradient to double-rounding (per-thread partial sums)."""
    from oracle import oracle_c
    from tests.helpers import scene_kwargs
    sc = make_scene("random1k", n_views=3, P=700, use_rgb=use_rgb, blend=blend)
    kw, bl = scene_kwargs(sc)
    dimg = dimg_like(3, sc.H, sc.W)
    res = []
    for par in (False, True):
        oracle_c.set_parallel(par)
        try:
            oracle_c.timing(reset=True)
            o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=True, **kw, **bl)
            g = o.backward(dimg)
            tot, ser = oracle_c.timing()
            assert tot > 0 and 0 <= ser <= tot
            res.append((o.image.clone(), {k: v.clone() for k, v in o.debug.items()}, g, ser / tot))
            o.close()
        finally:
            oracle_c.set_parallel(False)
    (img0, d0, g0, f0), (img1, d1, g1, f1) = res
    assert torch.equal(img0, img1)
    for k in ("sorted_keys", "sorted_gid", "ranges", "final_T", "n_contrib"):
        assert torch.equal(d0[k], d1[k]), k
    for k in g0:
        if k == "color_w" and g0[k].numel() == 48:
            assert rel_l2(g1[k], g0[k]) < 1e-6
        else:
            assert torch.equal(g0[k], g1[k]), k
    assert f1 <= f0                                   # the baseline mode has less single-threaded time, never more


def test_baseline_mode_with_a_smaller_team_than_asked_for():
    """the baseline mode partitions the Gaussians into omp_get_max_threads() slices; a runtime that delivers FEWER
    threads (OMP_THREAD_LIMIT, OMP_DYNAMIC, a cgroup quota) must still process every slice — the published cpu_baseline rests on
    this path. A child process with OMP_NUM_THREADS=8 and OMP_THREAD_LIMIT=3: parallel mode == serial mode bit for bit."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import torch\n"
        "from guassianhand_amd.scenes import make_scene\n"
        "from oracle import oracle_c\n"
        "from oracle.oracle_c import OracleRender\n"
        "sc = make_scene('random1k', n_views=2, P=900)\n"
        "res = []\n"
        "for par in (False, True):\n"
        "    oracle_c.set_parallel(par)\n"
        "    o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=True, colors_precomp=sc.shs.squeeze(1))\n"
        "    res.append((o.image.clone(), o.debug['sorted_keys'].clone(), o.debug['sorted_gid'].clone(), int(o.num_rendered)))\n"
        "    o.close()\n"
        "oracle_c.set_parallel(False)\n"
        "assert res[0][3] == res[1][3] and res[0][3] > 1000\n"
        "assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])\n"
        "print('ok', oracle_c.num_threads())\n")
    env = dict(os.environ, OMP_NUM_THREADS="8", OMP_THREAD_LIMIT="3", OMP_DYNAMIC="false", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


def test_c_oracle_against_the_float64_dense_oracle_over_the_feature_matrix():
    """tools/fuzz_oracles_cpu.py, 250 fixed-seed draws: the C oracle (what the HIP path is bit-equal to) against Oracle A (dense float64
    autograd from the behavioural spec) over RGB / SH degree 0-3 x M x every blend subset and form x off-grid sizes — the checker's
    formulas, not only its typing. (6,000 draws: profiles/r4_fuzz_oracles_cpu.txt.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_oracles_cpu.py"), "250", "3"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("oracle-vs-oracle fuzz")]
    assert line and ": 0 findings" in line[0], r.stdout[-2000:]
