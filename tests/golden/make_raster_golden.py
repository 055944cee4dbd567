#!/usr/bin/env python
"""Generate tests/golden/raster_golden.npz: small seeded input/output vectors for the rasteriser boundary.

The reference holds no golden vectors for this path (its rasteriser is an un-vendored CUDA dependency,
environment.yml:129), so these are produced by the repo's own CPU oracle (oracle/gh_oracle.c) AFTER it has
been pinned by the known-answer tests and the independent autograd oracle. They freeze the oracle against
regressions (CPU test) and are what the HIP path is compared with on the GPU box (GPU test).

Usage: python tests/golden/make_raster_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from guassianhand_amd.camera import intrinsics, look_at_w2c, pack_cameras_from_w2c  # noqa: E402
from oracle.oracle_c import OracleRender  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raster_golden.npz")
CASES = [dict(name="rgb", P=300, H=48, W=70, rgb=True, blend=False, nv=1, seed=1),
         dict(name="sh3_blend", P=260, H=40, W=56, rgb=False, blend=True, nv=2, seed=2),
         dict(name="rgb_blend_wpg", P=200, H=33, W=47, rgb=True, blend="wpg", nv=2, seed=3)]


def make_case(c):
    g = torch.Generator().manual_seed(c["seed"])
    P, H, W = c["P"], c["H"], c["W"]
    t = dict(means3D=torch.rand(P, 3, generator=g) * 0.2 - 0.1,
             opacities=torch.sigmoid(1.5 * torch.randn(P, generator=g)),
             scales=torch.exp(-4.6 + 0.4 * torch.randn(P, 3, generator=g)),
             rotations=torch.nn.functional.normalize(torch.randn(P, 4, generator=g)))
    if c["rgb"]:
        t["colors_precomp"] = torch.rand(P, 3, generator=g)
    else:
        t["shs"] = 0.3 * torch.randn(P, 16, 3, generator=g)
        t["shs"][:, 0] += 0.5
    if c["blend"]:
        t["color_w"] = 1 + 0.05 * torch.randn((P, 48) if c["blend"] == "wpg" else (48,), generator=g)
        t["color_b"] = 0.02 * torch.randn(P, 48, generator=g)
        t["opacity_b"] = 0.02 * torch.randn(P, generator=g)
        t["xyz_b"] = 0.004 * torch.randn(3, generator=g)
    w2cs, Ks = [], []
    for v in range(c["nv"]):
        eye = [0.3 * v - 0.1, 0.05 * v, -0.8]
        w2cs.append(look_at_w2c(eye, [0.0, 0.0, 0.0]))
        Ks.append(intrinsics(0.9 * W, W / 2.0 + 1.5, H / 2.0 - 0.75, skew=0.3 * v))
    t["cams"] = pack_cameras_from_w2c(torch.stack(w2cs), torch.stack(Ks), H, W, torch.tensor([0.1, 0.2, 0.3]))
    t["dL_dimage"] = torch.randn(c["nv"], 3, H, W, generator=g)
    return t


def run_oracle(c, t):
    kw = {k: t[k] for k in ("colors_precomp", "shs", "xyz_b", "opacity_b", "color_w", "color_b") if k in t}
    if not c["rgb"]:
        kw["sh_degree"] = 3
    o = OracleRender(t["cams"], t["means3D"], t["opacities"], t["scales"], t["rotations"], H=c["H"], W=c["W"], debug=True, **kw)
    return o, o.backward(t["dL_dimage"])


def main():
    out = {"cases": np.array([c["name"] for c in CASES])}
    for c in CASES:
        t = make_case(c)
        o, grads = run_oracle(c, t)
        n = c["name"]
        out[f"{n}_HW"] = np.array([c["H"], c["W"]])
        for k, v in t.items():
            out[f"{n}_in_{k}"] = v.numpy()
        out[f"{n}_out_image"] = o.image.numpy()
        out[f"{n}_out_radii"] = o.radii.numpy()
        out[f"{n}_out_num_rendered"] = np.array(o.num_rendered)
        out[f"{n}_out_n_contrib"] = o.debug["n_contrib"].numpy().astype(np.int32)
        for k, v in grads.items():
            out[f"{n}_grad_{k}"] = v.numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
