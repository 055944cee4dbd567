"""Known-answer tests that pin the CPU oracles (SURVEY.md §8c i-vii). The reference holds no golden
vectors for this path (parity unpinned), so the oracle is pinned by closed-form results of the published
algorithm; every parity-critical constant of Appendix A gets its own case. Both Oracle B (C, the checker
used by the GPU parity tests) and Oracle A (dense PyTorch) must reproduce them."""
import math

import pytest
import torch

from guassianhand_amd.camera import intrinsics, pack_cameras_from_w2c
from oracle import oracle_torch as OT
from oracle.oracle_c import OracleRender, gho_exp

H = W = 32
F_PX = 100.0


def cam(bg=(0.0, 0.0, 0.0), cx=8.5, cy=8.5):
    # identity w2c: camera at the origin looking down +z; a point on the optical axis lands on pixel
    # (cx-0.5, cy-0.5) because px = fx*x/z + cx - 0.5 (App. A.1-7: ((ndc+1)*W-1)/2).
    K = intrinsics(F_PX, cx, cy)
    return pack_cameras_from_w2c(torch.eye(4)[None], K[None], H, W, torch.tensor(bg))


def iso_radius(var):
    """3-sigma radius of an isotropic footprint of variance `var`: the eigenvalue discriminant is floored
    at 0.1 (App. A.1-6), so lambda_max = var + sqrt(0.1) even for a perfect circle."""
    return math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))


def gaussians(xyz, opac, scale, rgb):
    P = len(xyz)
    return dict(means3D=torch.tensor(xyz, dtype=torch.float32), opacities=torch.tensor(opac, dtype=torch.float32).reshape(P, 1),
                scales=torch.tensor(scale, dtype=torch.float32).reshape(P, 1).expand(P, 3).contiguous(),
                rotations=torch.tensor([[1.0, 0, 0, 0]] * P), colors=torch.tensor(rgb, dtype=torch.float32))


def run_both(g, c):
    o = OracleRender(c, g["means3D"], g["opacities"], g["scales"], g["rotations"], H=H, W=W,
                     colors_precomp=g["colors"], debug=True)
    r = c[0]
    img, radii, aux = OT.rasterize_dense(g["means3D"], g["opacities"], g["scales"], g["rotations"],
                                         viewmatrix=r[:16].reshape(4, 4), projmatrix=r[16:32].reshape(4, 4),
                                         campos=r[32:35], tanfovx=float(r[35]), tanfovy=float(r[36]), bg=r[37:40],
                                         H=H, W=W, colors_precomp=g["colors"], return_aux=True)
    assert torch.equal(radii, o.radii[0])
    assert (img - o.image[0]).abs().max() < 2e-6
    assert torch.equal(aux["n_contrib"].int(), o.debug["n_contrib"][0])
    return o, img


def test_reproducible_exp_matches_libm():
    for x in [0.0, -1e-6, -0.1, -0.5, -1.0, -2.5, -5.54, -10.0, -40.0, -87.0]:
        assert gho_exp(x) == pytest.approx(math.exp(x), rel=4e-7)
    assert gho_exp(-200.0) == 0.0


def test_i_single_isotropic_gaussian_closed_form():
    s, z, o, col, bg = 0.02, 1.0, 0.6, (0.9, 0.5, 0.1), (0.2, 0.3, 0.4)
    orc, _ = run_both(gaussians([[0, 0, z]], [o], [s], [col]), cam(bg))
    var = (F_PX * s / z) ** 2 + 0.3            # EWA footprint + 0.3 dilation (A.1-4)
    img = orc.image[0]
    for (dx, dy) in [(0, 0), (1, 0), (0, 2), (2, 1), (3, 3)]:
        a = o * math.exp(-(dx * dx + dy * dy) / (2 * var))
        for ch in range(3):
            want = col[ch] * a + (1 - a) * bg[ch] if a >= 1 / 255 else bg[ch]
            assert float(img[ch, 8 + dy, 8 + dx]) == pytest.approx(want, abs=2e-6)
    assert float(orc.debug["xy"][0, 0, 0]) == pytest.approx(8.0, abs=1e-5)
    assert int(orc.radii[0, 0]) == iso_radius(var)


def test_ii_two_gaussians_front_to_back():
    g = gaussians([[0, 0, 1.2], [0, 0, 1.0]], [0.5, 0.4], [0.02, 0.02], [(1, 0, 0), (0, 1, 0)])
    orc, _ = run_both(g, cam())
    px = orc.image[0][:, 8, 8]
    # index 1 is nearer: blended first
    assert float(px[1]) == pytest.approx(0.4, abs=1e-6)
    assert float(px[0]) == pytest.approx(0.5 * (1 - 0.4), abs=1e-6)
    assert float(orc.debug["final_T"][0, 8, 8]) == pytest.approx(0.6 * 0.5, abs=1e-6)


def test_iii_alpha_clamped_at_099():
    orc, _ = run_both(gaussians([[0, 0, 1.0]], [1.0], [0.02], [(1, 1, 1)]), cam((0.5, 0.5, 0.5)))
    assert float(orc.image[0][0, 8, 8]) == pytest.approx(0.99 + 0.01 * 0.5, abs=1e-6)
    # opacity > 1 after `+ opacity_b` (renderer_one_shot.py:306-307) is clamped the same way
    orc2, _ = run_both(gaussians([[0, 0, 1.0]], [1.3], [0.02], [(1, 1, 1)]), cam((0.5, 0.5, 0.5)))
    assert float(orc2.image[0][0, 8, 8]) == pytest.approx(0.99 + 0.01 * 0.5, abs=1e-6)
    # negative opacity never contributes
    orc3, _ = run_both(gaussians([[0, 0, 1.0]], [-0.2], [0.02], [(1, 1, 1)]), cam((0.5, 0.5, 0.5)))
    assert float(orc3.image[0][0, 8, 8]) == pytest.approx(0.5, abs=1e-7)


def test_iv_near_plane_cull_at_02():
    g = gaussians([[0, 0, 0.19], [0, 0, 0.2], [0, 0, 0.21]], [0.5] * 3, [0.001] * 3, [(1, 1, 1)] * 3)
    orc, _ = run_both(g, cam())
    assert orc.radii[0].tolist()[0] == 0 and orc.radii[0].tolist()[1] == 0 and orc.radii[0].tolist()[2] > 0


def test_v_early_stop_and_n_contrib():
    n = 10000
    g = gaussians([[0, 0, 1.0 + 1e-4 * i] for i in range(n)], [0.5] * n, [0.02] * n, [(1, 1, 1)] * n)
    orc, _ = run_both(g, cam())
    # T after k = 0.5^k ; the k-th is dropped when T*(1-a) < 1e-4 : 0.5^14 < 1e-4 <= 0.5^13
    assert int(orc.debug["n_contrib"][0, 8, 8]) == 13
    assert float(orc.debug["final_T"][0, 8, 8]) == pytest.approx(0.5 ** 13, rel=1e-6)
    assert float(orc.image[0][0, 8, 8]) == pytest.approx(1 - 0.5 ** 13, rel=1e-6)


def test_vi_equal_depth_ties_resolve_by_index():
    g = gaussians([[0, 0, 1.0], [0, 0, 1.0]], [0.5, 0.5], [0.02, 0.02], [(1, 0, 0), (0, 1, 0)])
    orc, _ = run_both(g, cam())
    px = orc.image[0][:, 8, 8]
    assert float(px[0]) == pytest.approx(0.5, abs=1e-6) and float(px[1]) == pytest.approx(0.25, abs=1e-6)
    assert orc.debug["sorted_gid"].tolist()[:2] == [0, 1]


def test_vii_subpixel_gaussian_survives_through_dilation():
    orc, _ = run_both(gaussians([[0, 0, 1.0]], [0.8], [1e-6], [(1, 1, 1)]), cam())
    assert float(orc.image[0][0, 8, 8]) == pytest.approx(0.8, abs=1e-5)
    assert float(orc.image[0][0, 8, 9]) == pytest.approx(0.8 * math.exp(-1 / 0.6), abs=1e-5)
    assert int(orc.radii[0, 0]) == iso_radius(0.3) == 3   # ceil(3*sqrt(0.3)) would be 2: the 0.1 floor matters


def test_pixel_centre_has_no_half_offset_and_tiles_bound_membership():
    # a Gaussian whose 3-sigma rect stops at the tile edge must not leak into the neighbouring tile even
    # where its alpha would still be >= 1/255 (tile membership is part of the result, App. A.4-1)
    s, o = 0.012, 0.9
    g = gaussians([[0.0, 0, 1.0]], [o], [s], [(1, 1, 1)])
    orc, _ = run_both(g, cam(cx=8.5, cy=8.5))
    var = (F_PX * s) ** 2 + 0.3
    r = iso_radius(var)
    assert 8 + r < 16 <= 8 + r + 15, "test geometry: rect must end inside tile 0"
    a_edge = o * math.exp(-((8 + r) - 8) ** 2 / (2 * var))
    assert float(orc.image[0][0, 8, 8 + r]) == pytest.approx(a_edge if a_edge >= 1 / 255 else 0.0, abs=1e-6)
    assert float(orc.image[0][0, 8, 16]) == 0.0 and float(orc.image[0][0, 16, 8]) == 0.0
