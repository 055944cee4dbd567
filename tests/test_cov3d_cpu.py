"""cov3D_precomp of the published module API (the reference never passes it, renderer_one_shot.py:313, :346): oracle side.
Oracle B (C, hand-written chain rule ending at dL/dSigma) against Oracle A (dense float64 autograd) and against the
scales + rotations path it must reproduce when Sigma = R S S^T R^T is handed over precomputed."""
import pytest
import torch

from guassianhand_amd.scenes import make_scene
from oracle import oracle_torch as OT
from oracle.oracle_c import OracleRender
from tests.helpers import dimg_like, rel_l2


def _sigma6(sc, mod=1.0):
    R = OT.quat_to_rot(sc.rotation.double())
    M = R * (mod * sc.scaling.double())[:, None, :]
    S = M @ M.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def test_cov3d_precomp_oracle_b_against_oracle_a_and_the_scale_rotation_path():
    sc = make_scene("random1k", n_views=2, P=400)
    cov = _sigma6(sc).float()
    cams = sc.cams()
    cols = sc.shs.squeeze(1)
    dimg = dimg_like(2, sc.H, sc.W)
    a = OracleRender(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, colors_precomp=cols)
    b = OracleRender(cams, sc.xyz, sc.opacity, None, None, H=sc.H, W=sc.W, colors_precomp=cols, cov3D_precomp=cov)
    # the same picture up to the float32 rounding of Sigma computed outside (float64 -> float32) instead of inside
    assert (a.image - b.image).abs().max().item() <= 2e-5
    assert int((a.radii != b.radii).sum()) <= 2
    gb = b.backward(dimg)
    assert set(gb) >= {"means3D", "opacities", "cov3D_precomp", "colors_precomp"} and "scales" not in gb and "rotations" not in gb
    # Oracle A, float64 autograd through the SAME inputs
    d = torch.float64
    leaves = dict(xyz=sc.xyz.to(d).requires_grad_(True), op=sc.opacity.to(d).requires_grad_(True), cov=cov.to(d).requires_grad_(True),
                  col=cols.to(d).requires_grad_(True))
    tot = 0
    for v in range(2):
        c = cams[v].to(d)
        img, _ = OT.rasterize_dense(leaves["xyz"], leaves["op"], None, None, viewmatrix=c[:16].reshape(4, 4), projmatrix=c[16:32].reshape(4, 4),
                                    campos=c[32:35], tanfovx=float(c[35]), tanfovy=float(c[36]), bg=c[37:40], H=sc.H, W=sc.W,
                                    colors_precomp=leaves["col"], cov3D_precomp=leaves["cov"])
        assert (img.float() - b.image[v]).abs().max().item() <= 1e-4
        tot = tot + (img * dimg[v].to(d)).sum()
    tot.backward()
    for k, ga in (("means3D", leaves["xyz"].grad), ("opacities", leaves["op"].grad.reshape(-1)), ("cov3D_precomp", leaves["cov"].grad),
                  ("colors_precomp", leaves["col"].grad)):
        assert rel_l2(gb[k].reshape(ga.shape), ga) <= 2e-4, (k, rel_l2(gb[k].reshape(ga.shape), ga))
    a.close(); b.close()


def test_exactly_one_of_scale_rotation_or_covariance():
    sc = make_scene("random1k", n_views=1, P=50)
    cov = _sigma6(sc).float()
    cols = sc.shs.squeeze(1)
    for scales, rots, c in ((sc.scaling, sc.rotation, cov), (None, None, None), (sc.scaling, None, None), (None, sc.rotation, cov)):
        with pytest.raises(RuntimeError, match="GH_ERR_INVALID_ARG"):
            OracleRender(sc.cams(), sc.xyz, sc.opacity, scales, rots, H=sc.H, W=sc.W, colors_precomp=cols, cov3D_precomp=c)
