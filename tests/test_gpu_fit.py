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
    fg = F.OneShotFit(pb_g["gs"], pb_g["uv"], map_hw=pb_g["map_hw"])
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
    true = F.OneShotFit(gs, uv, map_hw=map_hw)
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
    assert all(torch.isfinite(p).all() for p in (f.color_w, f.color_b_map, f.opacity_b_map))
