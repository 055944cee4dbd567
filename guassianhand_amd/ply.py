"""PLY export / import of a GaussianModel in the reference's attribute order.

Mirrors GaussianModel.construct_list_of_attributes / save_ply (tgs/models/renderer_one_shot.py:121-154):
properties  x y z nx ny nz  f_dc_*  f_rest_*  opacity  scale_*  rot_*  (all float32, binary little endian, one
`vertex` element — what plyfile's default `PlyData([el]).write(path)` emits), with
  * normals = 0,
  * f_dc   = shs[:, :1] flattened, f_rest = shs[:, 1:] flattened in (coefficient, channel) order — the reference
    does NOT transpose to channel-major like the original 3DGS exporter, and neither does this,
  * opacity stored as logit(clamp(opacity, 1e-3, 1-1e-3)), scale as log(scaling), rotation as given.
Written with numpy only (the reference depends on the third-party `plyfile`).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from .renderer import GaussianModel


def construct_list_of_attributes(gs: GaussianModel) -> List[str]:
    """renderer_one_shot.py:121-134."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    n_dc = gs.shs[:, :1].shape[1] * gs.shs.shape[2]
    n_rest = gs.shs[:, 1:].shape[1] * gs.shs.shape[2]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(gs.scaling.shape[1])] + [f"rot_{i}" for i in range(gs.rotation.shape[1])]
    return names


def save_ply(gs: GaussianModel, path: str) -> None:
    """renderer_one_shot.py:136-154."""
    xyz = gs.xyz.detach().cpu().numpy().astype(np.float32)
    normals = np.zeros_like(xyz)
    f_dc = gs.shs[:, :1].detach().flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = gs.shs[:, 1:].detach().flatten(start_dim=1).contiguous().cpu().numpy()
    op = torch.clamp(gs.opacity, 1e-3, 1 - 1e-3).detach().cpu().numpy()
    opacities = np.log(op / (1 - op))                                   # inverse_sigmoid (:24)
    scale = np.log(gs.scaling.detach().cpu().numpy())
    rotation = gs.rotation.detach().cpu().numpy()
    attributes = np.concatenate((xyz, normals, f_dc, f_rest, opacities, scale, rotation), axis=1).astype("<f4")
    names = construct_list_of_attributes(gs)
    assert attributes.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {xyz.shape[0]}\n" + \
             "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(attributes).tobytes())


def load_ply(path: str, device="cpu") -> GaussianModel:
    """Inverse of save_ply: opacity through sigmoid, scale through exp (so it round-trips up to the 1e-3 clamp)."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = f.readline().split()
        assert fmt[:2] == [b"format", b"binary_little_endian"], "only the reference's binary little-endian layout"
        n, names = 0, []
        while True:
            line = f.readline().strip()
            if line == b"end_header":
                break
            tok = line.split()
            if tok[0] == b"element":
                assert tok[1] == b"vertex"
                n = int(tok[2])
            elif tok[0] == b"property":
                assert tok[1] in (b"float", b"float32")
                names.append(tok[2].decode())
        data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    col = {k: i for i, k in enumerate(names)}
    pick = lambda prefix: np.stack([data[:, col[k]] for k in names if k.startswith(prefix)], axis=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    xyz = t(data[:, [col["x"], col["y"], col["z"]]])
    f_dc = pick("f_dc_")
    rest = [k for k in names if k.startswith("f_rest_")]
    shs = f_dc.reshape(n, 1, 3)
    if rest:
        shs = np.concatenate([shs, pick("f_rest_").reshape(n, -1, 3)], axis=1)
    return GaussianModel(xyz=xyz, opacity=torch.sigmoid(t(data[:, [col["opacity"]]])), rotation=t(pick("rot_")),
                         scaling=torch.exp(t(pick("scale_"))), shs=t(shs))
