"""The composed forward_single_batch (guassianhand_amd/renderer.py) against the reference's own
GS3DRenderer.forward_single_batch (renderer_one_shot.py:448-512), captured in the build container
(tests/golden/make_batch_fixture.py -> batch_fixture.npz): same stand-in sub-modules, same inputs."""
import os

import numpy as np
import pytest
import torch

from helpers import BatchStandIns, batch_inputs, forward_single_view

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "batch_fixture.npz"), allow_pickle=False)


def _run(dev, use_rgb, requires_grad=False):
    from guassianhand_amd.renderer import forward_single_batch
    st, inp = BatchStandIns(dev, use_rgb=use_rgb), batch_inputs()
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    if requires_grad:
        for k in ("feat", "pts", "color_w", "xyz_b", "color_b", "opacity_b"):
            d[k] = d[k].clone().requires_grad_(True)
    out = forward_single_batch(st.namespace(dev), d["feat"], d["pts"], d["w2cs"], d["Ks"], d["H"], d["W"], 0.71, 1.42, d["bg"],
                               color_w=d["color_w"], xyz_b=d["xyz_b"], color_b=d["color_b"], opacity_b=d["opacity_b"],
                               vert3d_uv=[None], face_uv=None, face_uv_xy=None)
    return st, d, out


@pytest.mark.parametrize("tag,use_rgb", [("rgb", True), ("sh", False)])
def test_composed_path_equals_the_reference_protocol_on_the_oracle(dev, fx, tag, use_rgb):
    """Keys, shapes and the Gaussians are the reference's; every view's comp_rgb / comp_mask equals the C oracle run on the very
    tensors the reference handed to its rasteriser (captured per call: blended means / opacities / colours, camera settings)."""
    from oracle.oracle_c import OracleRender
    st, d, out = _run(dev, use_rgb)
    assert sorted(out.keys()) == list(fx[f"{tag}_keys"])
    nv, H, W = d["w2cs"].shape[0], d["H"], d["W"]
    for k in ("comp_rgb", "comp_mask", "comp_rgb_bg"):
        assert tuple(out[k].shape) == tuple(fx[f"{tag}_shape_{k}"]), k
    assert np.array_equal(out["comp_rgb_bg"].cpu().numpy(), fx[f"{tag}_comp_rgb_bg"])
    for k in ("xyz", "opacity", "rotation", "scaling", "shs"):        # CPU vs GPU transcendental functions: a few ulps
        assert np.allclose(getattr(out["3dgs"], k).detach().cpu().numpy(), fx[f"{tag}_3dgs_{k}"], rtol=2e-6, atol=1e-7), k
    assert int(fx[f"{tag}_ncalls"]) == 2 * nv
    for v in range(nv):
        for ci, key in ((2 * v, "comp_rgb"), (2 * v + 1, "comp_mask")):
            t = lambda n: torch.tensor(fx[f"{tag}_call{ci}_{n}"])
            kw = dict(colors_precomp=t("colors_precomp")) if f"{tag}_call{ci}_colors_precomp" in fx.files else \
                dict(shs=t("shs"), sh_degree=int(fx[f"{tag}_call{ci}_sh_degree"]))
            orc = OracleRender(torch.tensor(fx[f"{tag}_call{ci}_cam"])[None], t("means3D"), t("opacities").reshape(-1), t("scales"),
                               t("rotations"), H=H, W=W, **kw)
            want = orc.image[0].permute(1, 2, 0)
            got = out[key][v].detach().cpu()
            assert (got - want).abs().max().item() <= 1e-4, (tag, v, key, (got - want).abs().max().item())
            orc.close()
    assert float(out["comp_mask"].max()) > 0.5 and float(out["comp_rgb"].std()) > 0.01      # something was drawn


@pytest.mark.parametrize("use_rgb", [True, False])
def test_composed_path_equals_the_two_call_protocol_with_gradients(dev, use_rgb):
    """The same composition written the reference's way on the same device — boolean-mask indexing, torch grid_sample lookups,
    forward_single_view's two GaussianRasterizer calls per view through the drop-in — images bit for bit, gradients of every
    differentiable input (features, points, blend parameters, UV maps) to rounding."""
    import torch.nn.functional as F
    from guassianhand_amd.camera import Camera
    st, d, out = _run(dev, use_rgb, requires_grad=True)
    gen = torch.Generator().manual_seed(3)
    w_rgb = torch.randn(out["comp_rgb"].shape, generator=gen).to(dev)
    w_msk = torch.randn(out["comp_mask"].shape, generator=gen).to(dev)
    leaves = ("feat", "pts", "color_w", "xyz_b", "color_b", "opacity_b")
    ((out["comp_rgb"] * w_rgb).sum() + (out["comp_mask"] * w_msk).sum()).backward()
    got = {k: d[k].grad.clone() for k in leaves}
    for k in leaves:
        d[k].grad = None
    # the reference's own statements (:468-510) over the same stand-ins
    s = st.gs_valid(d["feat"], d["pts"]).squeeze(1)
    pv, fv = d["pts"][s > st.threshold_low], d["feat"][s > st.threshold_low]
    pc, fc = d["pts"][s > st.threshold_high], d["feat"][s > st.threshold_high]
    pc = st.vert_pos_refinement(fc, pc)
    pts, feats = torch.cat([pv, pc], dim=-2), torch.cat([fv, fc], dim=-2)
    gs = st.forward_gs(feats, pts)
    uv, _, _ = st.get_uvd(pts, None, None, None)
    uv = uv.unsqueeze(0)
    uv[..., 0] = 2.0 * (uv[..., 0] / 1) - 1.0
    uv[..., 1] = 2.0 * (uv[..., 1] / 0.5) - 1.0
    look = lambda m: F.grid_sample(m[None], uv[:, :, None], align_corners=True, mode="bilinear").view(1, m.shape[0], -1).permute(0, 2, 1)[0]
    cb, ob = look(d["color_b"]), look(d["opacity_b"])
    rgb, msk = [], []
    for w2c, K in zip(d["w2cs"], d["Ks"]):
        r = forward_single_view(gs, Camera.from_w2c(w2c, K, d["H"], d["W"], 0.71, 1.42), d["bg"], color_w=d["color_w"], xyz_b=d["xyz_b"],
                                color_b=cb, opacity_b=ob, use_rgb=use_rgb)
        rgb.append(r["comp_rgb"]); msk.append(r["comp_mask"])
    rgb, msk = torch.stack(rgb), torch.stack(msk)
    assert torch.equal(gs.xyz, out["3dgs"].xyz) and torch.equal(gs.shs, out["3dgs"].shs)
    # (the kernel's bilinear weights and grid_sample's differ in the last bit, so "bit for bit" holds up to that lookup)
    assert (rgb - out["comp_rgb"]).abs().max().item() <= 2e-6 and (msk - out["comp_mask"]).abs().max().item() <= 2e-6
    ((rgb * w_rgb).sum() + (msk * w_msk).sum()).backward()
    for k in leaves:
        a, b = got[k], d[k].grad
        assert bool(torch.isfinite(a).all()) and float(b.abs().max()) > 0, k
        rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
        assert rel <= 1e-4, (k, rel)


@pytest.mark.parametrize("lo,hi,what", [(0.1, 2.0, "no row is duplicated"), (2.0, 3.0, "no row survives the prune")])
def test_composed_path_with_empty_selections(dev, lo, hi, what):
    """The two degenerate outcomes of the selection (renderer_one_shot.py:469-473): nothing above threshold_high (the refinement network
    sees zero rows) and nothing above threshold_low (zero Gaussians: every view is the background, the mask is zero)."""
    from guassianhand_amd.renderer import forward_single_batch
    st, inp = BatchStandIns(dev, use_rgb=True), batch_inputs(N=150)
    st.threshold_low, st.threshold_high = lo, hi
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    out = forward_single_batch(st.namespace(dev), d["feat"], d["pts"], d["w2cs"], d["Ks"], d["H"], d["W"], 0.71, 1.42, d["bg"],
                               color_w=d["color_w"], xyz_b=d["xyz_b"], color_b=d["color_b"], opacity_b=d["opacity_b"], vert3d_uv=[None])
    s = d["feat"][:, 0]
    n = int((s > lo).sum()) + int((s > hi).sum())
    assert out["3dgs"].xyz.shape[0] == n and out["comp_rgb"].shape == (2, d["H"], d["W"], 3), what
    if n == 0:
        assert torch.equal(out["comp_rgb"], d["bg"].expand(2, d["H"], d["W"], 3)) and float(out["comp_mask"].abs().max()) == 0.0
    else:
        assert float(out["comp_mask"].max()) > 0.5
