"""The committed golden vectors (tests/golden/raster_golden.npz) must be reproduced by the CPU oracle
bit for bit in the forward (integer + fp32 with a fixed arithmetic contract) and to rounding in the
backward. Guards the checker itself against regressions."""
import os

import numpy as np
import torch

from oracle.oracle_c import OracleRender
from tests.helpers import rel_l2


def load_case(z, n):
    t = {k[len(n) + 4:]: torch.tensor(z[k]) for k in z.files if k.startswith(n + "_in_")}
    H, W = [int(v) for v in z[f"{n}_HW"]]
    return t, H, W


def test_oracle_reproduces_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "raster_golden.npz"))
    for n in [str(c) for c in z["cases"]]:
        t, H, W = load_case(z, n)
        kw = {k: t[k] for k in ("colors_precomp", "shs", "xyz_b", "opacity_b", "color_w", "color_b") if k in t}
        if "shs" in t:
            kw["sh_degree"] = 3
        o = OracleRender(t["cams"], t["means3D"], t["opacities"], t["scales"], t["rotations"], H=H, W=W, debug=True, **kw)
        assert np.array_equal(o.image.numpy(), z[f"{n}_out_image"]), n
        assert np.array_equal(o.radii.numpy(), z[f"{n}_out_radii"])
        assert o.num_rendered == int(z[f"{n}_out_num_rendered"])
        assert np.array_equal(o.debug["n_contrib"].numpy(), z[f"{n}_out_n_contrib"])
        g = o.backward(t["dL_dimage"])
        for k, v in g.items():
            assert rel_l2(v, torch.tensor(z[f"{n}_grad_{k}"])) < 1e-6, (n, k)
