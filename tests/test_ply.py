"""PLY export in the reference's attribute order (renderer_one_shot.py:121-154), round trip, header layout."""
import numpy as np
import torch

from guassianhand_amd import ply
from guassianhand_amd.renderer import GaussianModel


def model(P, M, seed=0):
    g = torch.Generator().manual_seed(seed)
    return GaussianModel(xyz=torch.randn(P, 3, generator=g), opacity=torch.rand(P, 1, generator=g).clamp(0.01, 0.99),
                         rotation=torch.nn.functional.normalize(torch.randn(P, 4, generator=g)),
                         scaling=torch.exp(-5 + torch.randn(P, 3, generator=g)), shs=torch.randn(P, M, 3, generator=g))


def test_attribute_order_matches_reference():
    names = ply.construct_list_of_attributes(model(4, 16))
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"]
    assert names[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"] and names[9] == "f_rest_0" and names[53] == "f_rest_44"
    assert names[54:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert ply.construct_list_of_attributes(model(4, 1))[6:10] == ["f_dc_0", "f_dc_1", "f_dc_2", "opacity"]   # use_rgb: (P,1,3)


def test_round_trip_and_file_layout(tmp_path):
    for M in (1, 16):
        gs = model(257, M, seed=M)
        path = str(tmp_path / f"g{M}.ply")
        ply.save_ply(gs, path)
        raw = open(path, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 257\nproperty float x\n")
        ncol = 6 + 3 * M + 1 + 3 + 4
        assert len(body) == 257 * ncol * 4
        rows = np.frombuffer(body, dtype="<f4").reshape(257, ncol)
        assert np.array_equal(rows[:, 3:6], np.zeros((257, 3), np.float32))                       # normals are zero
        assert np.allclose(rows[:, 6:9], gs.shs[:, 0].numpy())                                    # f_dc = coefficient 0
        if M > 1:
            assert np.allclose(rows[:, 9:12], gs.shs[:, 1].numpy())                               # (coefficient, channel) order
        op = gs.opacity.numpy()
        assert np.allclose(rows[:, 6 + 3 * M], np.log(op / (1 - op))[:, 0], rtol=1e-5, atol=1e-6)  # logit
        assert np.allclose(rows[:, 7 + 3 * M:10 + 3 * M], np.log(gs.scaling.numpy()), rtol=1e-6)   # log scale
        back = ply.load_ply(path)
        for a, b in zip(gs, back):
            assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_opacity_is_clamped_like_the_reference(tmp_path):
    gs = model(8, 1)._replace(opacity=torch.tensor([[0.0], [1.0], [0.5], [1e-5], [0.9999], [0.3], [0.7], [0.2]]))
    path = str(tmp_path / "c.ply")
    ply.save_ply(gs, path)
    back = ply.load_ply(path)
    assert torch.allclose(back.opacity[:2, 0], torch.tensor([1e-3, 1 - 1e-3]), atol=1e-6)
