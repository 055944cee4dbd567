"""One-shot fit loop on the HIP renderer: the first step's parameter gradients equal those of the
same loop driven by the dense autograd oracle, and the loop fits the target at BASELINE configs[3]'s shape
(8 novel-view cameras)."""
import pytest
import torch

from tests.helpers import max_rel, oracle_render_views, rel_l2, tiny_fit_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def test_first_step_gradients_match_oracle_driven_loop(dev):
    from guassianhand_amd import fit as F
    pb_c = tiny_fit_problem()
    pb_g = tiny_fit_problem(device=dev)
    fc = F.OneShotFit(pb_c["gs"], pb_c["uv"], map_hw=pb_c["map_hw"], render_fn=oracle_render_views)
    fg = F.OneShotFit(pb_g["gs"], pb_g["uv"], map_hw=pb_g["map_hw"], active_texels=False)   # dense maps: arbitrary start values
    with torch.no_grad():
        for f, pb in ((fc, pb_c), (fg, pb_g)):
            f.color_w.copy_(1 + 0.5 * (pb["true"]["color_w"] - 1)); f.color_b.copy_(0.5 * pb["true"]["color_b"])
            f.opacity_b.copy_(0.5 * pb["true"]["opacity_b"])
        tgt = F.OneShotFit(pb_c["gs"], pb_c["uv"], map_hw=pb_c["map_hw"], render_fn=oracle_render_views)
        tgt.color_w.copy_(pb_c["true"]["color_w"]); tgt.color_b.copy_(pb_c["true"]["color_b"]); tgt.opacity_b.copy_(pb_c["true"]["opacity_b"])
        out = tgt.render(pb_c["w2c"], pb_c["K"], pb_c["H"], pb_c["W"], pb_c["bg"], tgt.blend_values())
        gt_rgb, gt_mask = out["comp_rgb"], out["comp_mask"].mean(-1)
    lc = fc.step(pb_c["w2c"], pb_c["K"], pb_c["H"], pb_c["W"], pb_c["bg"], gt_rgb, gt_mask)
    lg = fg.step(pb_g["w2c"], pb_g["K"], pb_g["H"], pb_g["W"], pb_g["bg"], gt_rgb.to(dev), gt_mask.to(dev))
    assert float(lg) == pytest.approx(float(lc), rel=2e-5)
    for n in ("color_w", "color_b_map", "opacity_b_map"):
        a, b = getattr(fg, n).grad.cpu(), getattr(fc, n).grad
        assert rel_l2(a, b) <= 2e-4 and max_rel(a, b, floor=1e-2) <= 5e-3, n


def test_uv_sample_kernels_match_grid_sample(dev):
    """the device UV lookup (channel-last map, lanes = channels) equals F.grid_sample(bilinear,
    align_corners=True, zeros padding) of renderer_one_shot.py:435-440, forward and scatter-add backward, including
    coordinates on and beyond the map border."""
    import torch.nn.functional as Fn
    from guassianhand_amd.uvmap import to_channel_last, uv_sample
    g = torch.Generator().manual_seed(9)
    for C_, Hm, Wm, P in ((48, 37, 53, 5000), (1, 64, 128, 3000), (5, 8, 9, 777)):
        chw = torch.randn(C_, Hm, Wm, generator=g)
        uv = torch.rand(P, 2, generator=g) * 2.4 - 1.2            # some samples fall outside [-1,1]
        uv[:7] = torch.tensor([[-1, -1], [1, 1], [1, -1], [-1, 1], [0, 0], [1.0, 0.3], [-1.0, -0.7]])
        ref_in = chw.clone().requires_grad_(True)
        ref = Fn.grid_sample(ref_in[None], uv[None, :, None, :], align_corners=True, mode="bilinear")[0, :, :, 0].T
        dout = torch.randn(P, C_, generator=g)
        (ref * dout).sum().backward()
        m = to_channel_last(chw).to(dev).requires_grad_(True)
        out = uv_sample(m, uv.to(dev))
        (out * dout.to(dev)).sum().backward()
        assert torch.allclose(out.detach().cpu(), ref.detach(), atol=2e-6, rtol=1e-5)
        assert torch.allclose(m.grad.permute(2, 0, 1).cpu(), ref_in.grad, atol=2e-5, rtol=1e-4)


def test_fit_converges_on_eight_views(dev):
    """configs[3] shape: 8 ring cameras, two-hand Gaussians (reduced P), blend maps learned from the images."""
    from guassianhand_amd import fit as F
    from guassianhand_amd.renderer import GaussianModel
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("two_hands", n_views=8, P=20000, blend=False).to(dev)
    g = torch.Generator().manual_seed(4)
    uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
    gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
    map_hw = (64, 128)
    true = F.OneShotFit(gs, uv, map_hw=map_hw, active_texels=False)
    with torch.no_grad():
        true.color_w.copy_((1 + 0.1 * torch.randn(48, generator=g)).to(dev))
        true.color_b.copy_((0.1 * torch.randn(48, *map_hw, generator=g)).to(dev))          # writes through the layout view
        true.opacity_b.copy_((0.05 * torch.randn(1, *map_hw, generator=g)).to(dev))
        out = true.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, true.blend_values())
        gt_rgb, gt_mask = out["comp_rgb"].clone(), out["comp_mask"].mean(-1).clone()
    f = F.OneShotFit(gs, uv, map_hw=map_hw)
    losses = [float(f.step(sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask, sync=(i == 0))) for i in range(50)]   # 1 epoch x 50 steps
    from guassianhand_amd import rasterizer as R
    R.check_overflow()
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert f.active and all(torch.isfinite(p).all() for p in (f.color_w, f.color_b_tex, f.opacity_b_tex))


def test_active_texel_gather_is_bit_identical_to_dense_lookup(dev):
    """gh_uv_gather_forward over the compacted active texels == gh_uv_sample_forward over the dense map, bit for bit,
    including UVs on / outside the border; compact() / dense() round-trip; the backward scatters the same sums."""
    from guassianhand_amd.uvmap import ActiveTexels, uv_gather, uv_gather_backward, uv_sample
    g = torch.Generator().manual_seed(21)
    for C_, Hm, Wm, P in ((48, 37, 53, 4000), (1, 64, 128, 3000)):
        uv = torch.rand(P, 2, generator=g) * 2.4 - 1.2
        uv[:6] = torch.tensor([[-1, -1], [1, 1], [1, -1], [-1, 1], [0, 0], [1.0, 0.3]])
        uv = uv.to(dev)
        at = ActiveTexels(uv, Hm, Wm)
        assert 0 < at.U <= 4 * P and at.slot.shape == (P, 4) and int(at.slot.max()) == at.U - 1
        dense = torch.randn(Hm, Wm, C_, generator=g).to(dev)
        tex = at.compact(dense)
        masked = at.dense(tex)                                     # dense map with the inactive texels zeroed
        assert torch.equal(at.compact(masked), tex)
        assert torch.equal(uv_gather(tex, at), uv_sample(dense, uv))           # inactive texels are never read
        dout = torch.randn(P, C_, generator=g).to(dev)
        gt = torch.zeros_like(tex)
        uv_gather_backward(dout, at, gt)
        m = dense.clone().requires_grad_(True)
        (uv_sample(m, uv) * dout).sum().backward()
        assert torch.equal(at.dense(gt), m.grad)                    # both go through the same fixed-order gather
        assert float((m.grad - at.dense(at.compact(m.grad))).abs().max()) == 0.0    # no gradient outside the active set
        # the texel lists: every valid (Gaussian, corner) pair exactly once, grouped by its texel, ascending inside a texel
        flat = at.slot.reshape(-1).long().cpu()
        pairs, rp = at.pairs.long().cpu(), at.row_ptr.long().cpu()
        assert rp[0] == 0 and rp[-1] == pairs.numel() == int((flat >= 0).sum()) and rp.numel() == at.U + 1
        assert torch.equal(flat[pairs], torch.repeat_interleave(torch.arange(at.U), rp[1:] - rp[:-1]))
        assert torch.equal(torch.sort(pairs).values, torch.nonzero(flat >= 0).reshape(-1))
        seg_start = torch.zeros(pairs.numel(), dtype=torch.bool); seg_start[rp[:-1][rp[:-1] < pairs.numel()]] = True
        assert bool(((pairs[1:] > pairs[:-1]) | seg_start[1:]).all())
        # ... against the atomic entry point of the C-ABI (kept for callers without an index): same sums up to order noise,
        # and the sorted form is bitwise reproducible run to run
        import ctypes as C
        from guassianhand_amd import _lib
        ga = torch.zeros_like(tex)
        rc = _lib.lib().gh_uv_gather_backward(C.c_void_p(at.slot.data_ptr()), C.c_void_p(at.w.data_ptr()), C.c_void_p(dout.data_ptr()),
                                              C.c_void_p(ga.data_ptr()), P, C_, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0 and torch.allclose(ga, gt, atol=2e-5, rtol=1e-4)
        for _ in range(3):
            g2 = torch.zeros_like(tex)
            uv_gather_backward(dout, at, g2)
            assert torch.equal(g2, gt)


def test_fused_adam_regulariser_step_matches_torch_adam(dev):
    """gh_adam_reg_step == torch.optim.Adam on grad + d/dp (l1*sum|p| + l2*sum p^2), several steps, with an lr change;
    returns the regulariser sums of the pre-update values and clears the gradient buffer."""
    from guassianhand_amd.uvmap import AdamReg
    g = torch.Generator().manual_seed(3)
    n, l1, l2 = 100_003, 3e-4, 2e-3
    p0 = (0.1 * torch.randn(n, generator=g)).to(dev)
    p0[:100] = 0.0
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.01)
    mine = AdamReg(p0.clone(), 0.01, reg_l1=l1, reg_l2=l2)
    for it in range(6):
        gimg = (0.01 * torch.randn(n, generator=g)).to(dev)
        gimg[:50] = 0.0                                             # zero parameter + zero gradient must stay exactly 0
        lr = 0.01 if it < 3 else 0.005
        opt.param_groups[0]["lr"] = lr
        opt.zero_grad()
        reg = l1 * ref.abs().sum() + l2 * ref.pow(2).sum()
        reg.backward()
        ref.grad += gimg
        want_sums = torch.stack([ref.detach().abs().sum(), ref.detach().pow(2).sum()])
        opt.step()
        mine.lr = lr
        mine.grad += gimg
        sums = mine.step()
        assert torch.allclose(sums, want_sums, rtol=1e-5)
        assert float(mine.grad.abs().max()) == 0.0
        assert torch.allclose(mine.param, ref.detach(), rtol=2e-5, atol=2e-7), it
        assert float(mine.param[:50].abs().max()) == 0.0


def test_grouped_adam_and_two_map_kernels_equal_the_single_calls(dev):
    """The fit's fused launches — gh_adam_reg_step_group (three tensors in one launch), gh_uv_gather_forward2 /
    gh_uv_scatter_sorted2 (colour-bias and opacity-bias maps in one launch) and gh_reg_total (loss assembly) — produce exactly
    what the single-tensor / single-map entry points produce, incl. an unaligned tail, the device-side step count and a
    guarded (skipped) step."""
    from guassianhand_amd.uvmap import (ActiveTexels, AdamReg, adam_group_step, reg_total, uv_gather, uv_gather2, uv_gather_backward,
                                        uv_gather_backward2)
    g = torch.Generator().manual_seed(8)
    sizes = (48, 100_003, 359_653 * 3)
    mk = lambda: [AdamReg((0.1 * torch.randn(n, generator=torch.Generator().manual_seed(n))).to(dev), 0.01, reg_l1=1e-4 * (i == 1),
                          reg_l2=1e-3 * (i == 2)) for i, n in enumerate(sizes)]
    one, grp = mk(), mk()
    guard_ok = torch.zeros(4, dtype=torch.int32, device=dev)
    guard_bad = torch.tensor([0, 1, 0, 0], dtype=torch.int32, device=dev)
    for it in range(5):
        guard = guard_bad if it == 2 else guard_ok
        ext = (0.01 * torch.randn(sizes[0], generator=g)).to(dev)            # tensor 0 steps from a caller-supplied gradient buffer
        for a, b in zip(one, grp):
            gr = (0.01 * torch.randn(a.param.numel(), generator=g)).to(dev)
            a.grad += gr; b.grad += gr
        for i, a in enumerate(one):
            a.step(guard, grad=ext.clone() if i == 0 else None, sums=False)
        adam_group_step(grp, guard, [ext.clone(), None, None])
        for a, b in zip(one, grp):
            for x, y in ((a.param, b.param), (a.exp_avg, b.exp_avg), (a.exp_avg_sq, b.exp_avg_sq), (a.partials, b.partials), (a.grad, b.grad)):
                assert torch.equal(x, y), it
            assert a.step_state.tolist() == b.step_state.tolist() == [it + 1 - (it >= 2), 0]
    tot = reg_total(grp[1], 0, 2.5, grp[2], 1, 0.5, base=torch.tensor(3.0, device=dev))
    want = 2.5 * grp[1].partials[:, 0].double().sum() + 0.5 * grp[2].partials[:, 1].double().sum()
    assert float(tot[1]) == pytest.approx(float(want), rel=1e-6) and float(tot[0]) == pytest.approx(3.0 + float(want), rel=1e-6)
    # two maps at the same UVs
    P, Hm, Wm = 5000, 37, 53
    uv = (torch.rand(P, 2, generator=g) * 2.2 - 1.1).to(dev)
    at = ActiveTexels(uv, Hm, Wm)
    ta, tb = torch.randn(at.U, 3, generator=g).to(dev), torch.randn(at.U, 1, generator=g).to(dev)
    oa, ob = uv_gather2(ta, tb, at)
    assert torch.equal(oa, uv_gather(ta, at)) and torch.equal(ob, uv_gather(tb, at))
    da, db = torch.randn(P, 3, generator=g).to(dev), torch.randn(P, generator=g).to(dev)
    ga, gb, ga2, gb2 = torch.zeros_like(ta), torch.zeros_like(tb), torch.zeros_like(ta), torch.zeros_like(tb)
    uv_gather_backward(da, at, ga); uv_gather_backward(db.reshape(-1, 1), at, gb)
    uv_gather_backward2(da, db, at, ga2, gb2)
    assert torch.equal(ga, ga2) and torch.equal(gb, gb2)


def test_active_texel_fit_equals_dense_fit(dev):
    """The active-texel fit (compact texels, fused regulariser + Adam) follows the dense torch.optim.Adam fit of the
    reference (infer_one_shot.py:345-349, :489-524) step by step, across an lr milestone, and leaves every inactive
    texel at exactly zero."""
    from guassianhand_amd import fit as F
    pb = tiny_fit_problem(device=dev)
    tgt = F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"], active_texels=False)
    with torch.no_grad():
        tgt.color_w.copy_(pb["true"]["color_w"]); tgt.color_b.copy_(pb["true"]["color_b"]); tgt.opacity_b.copy_(pb["true"]["opacity_b"])
        out = tgt.render(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], tgt.blend_values())
        gt_rgb, gt_mask = out["comp_rgb"].clone(), out["comp_mask"].mean(-1).clone()
    fd = F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"], active_texels=False)
    fa = F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"])
    assert fa.active and not fd.active
    for epoch in range(3):
        for it in range(3):
            ld = fd.step(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
            la = fa.step(pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
            assert float(la) == pytest.approx(float(ld), rel=1e-4)
        fd.end_epoch(); fa.end_epoch()
    assert fd.opt.param_groups[0]["lr"] == pytest.approx(0.005)     # milestone 2 passed
    # Adam normalises by sqrt(v): early steps amplify rounding differences of tiny gradients, hence the absolute floor
    assert torch.allclose(fa.color_w, fd.color_w, rtol=1e-3, atol=2e-4)
    assert torch.allclose(fa.color_b, fd.color_b, rtol=1e-3, atol=5e-4)
    assert torch.allclose(fa.opacity_b, fd.opacity_b, rtol=1e-3, atol=5e-4)
    inactive = torch.ones(pb["map_hw"][0] * pb["map_hw"][1], dtype=torch.bool, device=dev)
    inactive[fa.texels.index] = False
    assert float(fd.color_b_map.view(-1, 48)[inactive].abs().max()) == 0.0      # the premise of the active-texel mode
    assert float(fd.opacity_b_map.view(-1, 1)[inactive].abs().max()) == 0.0
    # export / import round trip through the reference layout
    fb = F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"])
    fb.load_maps(fa.color_b, fa.opacity_b)
    assert torch.equal(fb.color_b_tex, fa.color_b_tex) and torch.equal(fb.opacity_b_tex, fa.opacity_b_tex)
    with pytest.raises(ValueError):
        fb.load_maps(torch.ones_like(fa.color_b), fa.opacity_b)


def test_fused_l1_loss_matches_torch(dev):
    """gh_l1_loss == (img - gt).abs().mean() and its autograd gradient, incl. exact zeros (sign(0) = 0), odd sizes and
    an upstream gradient factor."""
    from guassianhand_amd.loss import l1_mean_loss
    g = torch.Generator().manual_seed(8)
    for shape in ((8, 3, 512, 334), (1, 3, 7, 5), (3,), (2, 1025)):
        img = torch.randn(*shape, generator=g).to(dev)
        gt = torch.randn(*shape, generator=g).to(dev)
        flat = img.view(-1)
        flat[:: 7] = gt.view(-1)[:: 7]                         # exact ties
        a = img.clone().requires_grad_(True)
        b = img.clone().requires_grad_(True)
        la = l1_mean_loss(a, gt)
        lb = (b - gt).abs().mean()
        (2.5 * la).backward(); (2.5 * lb).backward()
        assert float(la) == pytest.approx(float(lb), rel=1e-5)
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)
        assert float(a.grad.view(-1)[0]) == 0.0


def test_fused_fit_loss_matches_torch_restatement(dev):
    """gh_fit_loss == fit.fit_loss (the torch restatement of utils.py:180-294 / infer_one_shot.py:497-510) in value and
    in the gradients w.r.t. the rasteriser outputs: L1 with sign(0) = 0, bbox zeroing, the clip(-0.001, 1) of the mask
    term (alpha above 1 and below -0.001 carry no gradient), per-view means summed and scaled."""
    from guassianhand_amd import fit as F
    from guassianhand_amd.loss import fit_image_loss
    g = torch.Generator().manual_seed(31)
    NV, H, W = 3, 37, 29
    img = torch.rand(NV, 3, H, W, generator=g).to(dev)
    alpha = (torch.rand(NV, H, W, generator=g) * 1.3 - 0.1).to(dev)               # spans both clip edges
    gt_rgb = torch.rand(NV, H, W, 3, generator=g).to(dev)
    gt_mask = (torch.rand(NV, H, W, generator=g) > 0.5).float().to(dev)
    img.view(-1)[::11] = gt_rgb.permute(0, 3, 1, 2).reshape(-1)[::11]             # exact ties
    for bbox in (None, (torch.rand(NV, H, W, generator=g) > 0.3).to(dev)):
        a1, b1 = img.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
        a2, b2 = img.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
        l1 = fit_image_loss(a1, b1, gt_rgb, gt_mask, None if bbox is None else bbox.float(), scale=1.0 / 8)
        l2 = F.fit_loss(a2.permute(0, 2, 3, 1), b2.unsqueeze(-1).expand(-1, -1, -1, 3), gt_rgb, gt_mask, bbox) / 8
        (3.0 * l1).backward(); (3.0 * l2).backward()
        assert float(l1) == pytest.approx(float(l2), rel=2e-6)
        assert torch.allclose(a1.grad, a2.grad, rtol=1e-5, atol=1e-9)
        assert torch.allclose(b1.grad, b2.grad, rtol=1e-5, atol=1e-9)


def test_device_side_overflow_guard_keeps_a_sync_free_fit_from_stepping_on_garbage(dev):
    """with the instance capacity forced too small, a sync-free step must NOT move the parameters:
    the loss kernel emits NaN + zero gradients and gh_adam_reg_step is a no-op that does not count the step (device-side
    GhCounters.overflow guard). check_overflow() then raises and grows the capacity; the re-run step equals the step a
    fit that never overflowed takes."""
    from guassianhand_amd import fit as F
    from guassianhand_amd import rasterizer as R
    pb = tiny_fit_problem(P=600, n_views=4, hw=(64, 64), device=dev)
    mk = lambda: F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"])
    g = torch.Generator().manual_seed(3)
    gt_rgb = torch.rand(4, 64, 64, 3, generator=g).to(dev)
    gt_mask = (torch.rand(4, 64, 64, generator=g) > 0.5).float().to(dev)
    args = (pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
    ref = mk()
    l_ref = float(ref.step(*args, sync=True))
    key = R.capacity_key(600, 4, 64, 64, split=R._split_policy is True)             # ("auto" does not split 64x64 views)
    good_cap = R._capacity[key]
    f = mk()
    before = {k: a.param.clone() for k, a in f._adam.items()}
    R.check_overflow()
    R._capacity[key] = 64                                          # far below D
    try:
        l_bad = f.step(*args, sync=False)
        assert torch.isnan(l_bad).item(), "an overflowed render must report a NaN loss, not a number"
        for k, a in f._adam.items():
            assert torch.equal(a.param, before[k]), k               # parameters untouched
            assert float(a.exp_avg.abs().max()) == 0.0 and float(a.exp_avg_sq.abs().max()) == 0.0
            assert int(a.step_state.max()) == 0                     # ... and the step was not counted
        with pytest.raises(R.GhOverflowError):
            R.check_overflow()
        assert R._capacity[key] > 64
    finally:
        R._capacity[key] = max(R._capacity.get(key, 0), good_cap)
    l_again = float(f.step(*args, sync=False))
    R.check_overflow()
    assert l_again == l_ref
    for k in f._adam:
        assert torch.equal(f._adam[k].param, ref._adam[k].param), k      # no atomics anywhere on the fit path: bit for bit
        assert int(f._adam[k].step_state.max()) == 1


def test_captured_fit_step_replays_the_eager_fit(dev):
    """fit.CapturedFitStep: the whole step as one HIP graph. Its replays must take exactly the steps the eager loop takes —
    parameters, Adam moments and losses — across a learning-rate milestone (re-capture) and with the step count of
    the bias correction living on the device (gh_adam_reg_step's step_state)."""
    from guassianhand_amd import fit as F
    from guassianhand_amd import rasterizer as R
    pb = tiny_fit_problem(P=600, n_views=4, hw=(64, 64), device=dev)
    mk = lambda: F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"])
    g = torch.Generator().manual_seed(5)
    gt_rgb = torch.rand(4, 64, 64, 3, generator=g).to(dev)
    gt_mask = (torch.rand(4, 64, 64, generator=g) > 0.5).float().to(dev)
    args = (pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
    steps_per_epoch, n_steps = 3, 12                                   # epochs 0..3: the milestone at epoch 2 halves the rate
    eager, losses_e = mk(), []
    for i in range(n_steps):
        losses_e.append(float(eager.step(*args, sync=(i == 0))))
        if (i + 1) % steps_per_epoch == 0:
            eager.end_epoch()
    R.check_overflow()
    f = mk()
    cap = f.captured(*args)                                            # two regular steps
    losses_c = []
    for i in range(2, n_steps):
        if i % steps_per_epoch == 0:
            f.end_epoch()
        losses_c.append(float(cap.replay()))
    cap.check()
    # no atomics anywhere on the fit path (the texel scatter is a fixed-order gather): the replays ARE the eager steps
    assert losses_c == losses_e[2:]
    for k in eager._adam:
        a, b = eager._adam[k], f._adam[k]
        for x, y in ((a.param, b.param), (a.exp_avg, b.exp_avg), (a.exp_avg_sq, b.exp_avg_sq)):
            assert torch.equal(x, y), k
        assert int(b.step_state[0]) == n_steps and int(b.step_state[1]) == 0
        assert int(a.step_state[0]) == n_steps


def test_a_milestone_on_a_dropped_cache_recaptures_a_refresh_not_a_build(dev):
    """Found by tools/fuzz_fit.py: when the static lists were dropped (GeometryCache.clear_all(): any overflow in the process does
    that) AND the epoch count crossed a learning-rate milestone before the next replay, CapturedFitStep re-captured first — over
    an empty cache, so the graph recorded the BUILD (a full forward) and every later replay rebuilt the lists instead of refreshing
    them (slower, and the gradients of a build and of a refresh differ in the last bit). The stale check now comes first: that
    replay runs eagerly (it rebuilds), then the refresh is captured — the replays are the eager fit's steps again, bit for bit."""
    from guassianhand_amd import fit as F
    from guassianhand_amd import rasterizer as R
    pb = tiny_fit_problem(P=600, n_views=4, hw=(64, 80), device=dev)
    mk = lambda: F.OneShotFit(pb["gs"], pb["uv"], map_hw=pb["map_hw"])
    g = torch.Generator().manual_seed(9)
    gt_rgb = torch.rand(4, 64, 80, 3, generator=g).to(dev)
    gt_mask = (torch.rand(4, 64, 80, generator=g) > 0.5).float().to(dev)
    args = (pb["w2c"], pb["K"], pb["H"], pb["W"], pb["bg"], gt_rgb, gt_mask)
    eager, f = mk(), mk()
    for i in range(2):
        eager.step(*args, sync=(i == 0))
    cap = f.captured(*args)
    for _ in range(3):
        eager.step(*args, sync=False)
        cap.replay()
    R.GeometryCache.clear_all()
    for fit_ in (eager, f):
        fit_.end_epoch()
        fit_.end_epoch()                                                # epoch 2 is a milestone: the rate halves
    losses_e, losses_c = [], []
    for _ in range(5):
        losses_e.append(float(eager.step(*args, sync=False)))
        losses_c.append(float(cap.replay()))
    cap.check()
    R.check_overflow()
    assert losses_c == losses_e
    assert f._geom_cache.builds == eager._geom_cache.builds == 2
    for k in eager._adam:
        for name in ("param", "exp_avg", "exp_avg_sq"):
            assert torch.equal(getattr(eager._adam[k], name), getattr(f._adam[k], name)), (k, name)
