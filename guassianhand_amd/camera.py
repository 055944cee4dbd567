"""Pin-hole camera records for the rasteriser — host-side mirror of the reference's `Camera`.

Mirrors tgs/models/renderer_one_shot.py:61-112 (getProjectionMatrix_refine, intrinsic_to_fov,
Camera.__init__/from_w2c) and the settings assembly at :276-294, batched over views and without
host synchronisation: everything stays a device tensor and is packed into the GH_CAM_FLOATS record
include/gh_raster.h describes.
"""
from __future__ import annotations

import os
import threading
from collections import OrderedDict
from typing import Sequence

import torch

from ._abi import GH_CAM_FLOATS


def projection_matrix_refine(K: torch.Tensor, H: int, W: int, znear: float = 0.01, zfar: float = 1000.0) -> torch.Tensor:
    """Batched restatement of getProjectionMatrix_refine (renderer_one_shot.py:61-81). K: (...,>=3,>=3)."""
    fx, fy = K[..., 0, 0], K[..., 1, 1]
    cx, cy, s = K[..., 0, 2], K[..., 1, 2], K[..., 0, 1]
    Pm = torch.zeros(K.shape[:-2] + (4, 4), dtype=K.dtype, device=K.device)
    Pm[..., 0, 0] = 2 * fx / W
    Pm[..., 0, 1] = 2 * s / W
    Pm[..., 0, 2] = -1 + 2 * (cx / W)
    Pm[..., 1, 1] = 2 * fy / H
    Pm[..., 1, 2] = -1 + 2 * (cy / H)
    Pm[..., 2, 2] = (zfar + znear) / (zfar - znear)
    Pm[..., 2, 3] = -1 * 2 * zfar * znear / (zfar - znear)
    Pm[..., 3, 2] = 1.0
    return Pm


class Camera:
    """Same attributes as the reference's Camera (renderer_one_shot.py:90-107); znear/zfar are forced
    to 0.01/1000 regardless of the arguments, exactly as :99-100 do."""

    def __init__(self, w2c, intrinsic, FoVx, FoVy, height, width, znear=None, zfar=None):
        self.FoVx, self.FoVy = FoVx, FoVy
        self.height, self.width = int(height), int(width)
        self.world_view_transform = w2c.transpose(0, 1)
        self.zfar, self.znear = 1000.0, 0.01
        self.projection_matrix = projection_matrix_refine(intrinsic, self.height, self.width, self.znear,
                                                          self.zfar).transpose(0, 1).to(w2c.device)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = self.world_view_transform.inverse()[3, :3]

    @staticmethod
    def from_w2c(w2c, intrinsic, height, width, znear=None, zfar=None) -> "Camera":
        fx, fy = intrinsic[0, 0], intrinsic[1, 1]
        w = torch.tensor(width, device=w2c.device)
        h = torch.tensor(height, device=w2c.device)
        FoVx = 2 * torch.arctan2(w, 2 * fx)
        FoVy = 2 * torch.arctan2(h, 2 * fy)
        return Camera(w2c, intrinsic, FoVx, FoVy, height, width, znear, zfar)


_PACK_ON = os.environ.get("GH_PACK_CACHE", "1") != "0"     # (measurement A/B only)
_PACK_MAX = 64          # records kept (a step of the reference protocol cycles through 2 x its views: RGB call + mask call each)
_pack_lock = threading.Lock()
_pack_cache: "OrderedDict" = OrderedDict()      # key (tensor ids, versions, tan) -> (the four tensors: strong refs keep the ids valid, record)


def pack_camera(viewmatrix, projmatrix, campos, tanfovx, tanfovy, bg) -> torch.Tensor:
    """One GH_CAM_FLOATS record from the 12-field settings of the reference call (:281-294), built on the device without a
    host->device copy (a pageable copy would synchronise). A loop that hands over the very same, unmodified tensor objects again
    (the same settings objects step after step; the reference's RGB and mask call of a view: same camera tensors, another bg,
    :281-296, :355-370) gets the record it got before: a small bounded cache keyed by tensor identity + version, behind a lock
    (round 6: it was a lock-less single entry that the alternating bg of the two calls evicted every time)."""
    dev = viewmatrix.device
    tx, ty = float(tanfovx), float(tanfovy)
    key = (id(viewmatrix), id(projmatrix), id(campos), id(bg), viewmatrix._version, projmatrix._version, campos._version, bg._version, tx, ty)
    with _pack_lock:
        hit = _pack_cache.get(key) if _PACK_ON else None
        if hit is not None and hit[0] is viewmatrix and hit[1] is projmatrix and hit[2] is campos and hit[3] is bg:
            _pack_cache.move_to_end(key)
            return hit[4]
    tf = torch.empty(2, dtype=torch.float32, device=dev)
    tf[0].fill_(tx)
    tf[1].fill_(ty)
    b3 = bg if (bg.dtype is torch.float32 and bg.device == dev and bg.dim() == 1) else bg.reshape(3).float().to(dev)
    rec = torch.cat([viewmatrix.reshape(16).float(), projmatrix.reshape(16).float(), campos.reshape(3).float(), tf, b3]).reshape(1, GH_CAM_FLOATS)
    with _pack_lock:
        _pack_cache[key] = (viewmatrix, projmatrix, campos, bg, rec)
        while len(_pack_cache) > _PACK_MAX:
            _pack_cache.popitem(last=False)
    return rec


def clear_pack_cache() -> None:
    with _pack_lock:
        _pack_cache.clear()


def pack_cameras_from_w2c(w2cs: torch.Tensor, Ks: torch.Tensor, H: int, W: int, bg: torch.Tensor) -> torch.Tensor:
    """(Nv,4,4) w2c + (Nv,>=3,>=3) intrinsics -> (Nv, GH_CAM_FLOATS) records, no host sync.

    Per view this equals Camera.from_w2c + the settings of renderer_one_shot.py:276-294:
    tanfov = tan(atan2(size, 2 f)) = size / (2 f)  (evaluated through atan2/tan like the reference).
    """
    Nv = w2cs.shape[0]
    w2cs = w2cs.float()
    Ks = Ks.float()
    view = w2cs.transpose(1, 2)
    proj = projection_matrix_refine(Ks, H, W).transpose(1, 2)
    full = view @ proj
    campos = torch.linalg.inv(view)[:, 3, :3]
    fovx = 2 * torch.arctan2(torch.full_like(Ks[:, 0, 0], float(W)), 2 * Ks[:, 0, 0])
    fovy = 2 * torch.arctan2(torch.full_like(Ks[:, 1, 1], float(H)), 2 * Ks[:, 1, 1])
    tanx, tany = torch.tan(fovx * 0.5), torch.tan(fovy * 0.5)
    bgv = bg.float().reshape(-1, 3).expand(Nv, 3)
    return torch.cat([view.reshape(Nv, 16), full.reshape(Nv, 16), campos, tanx[:, None], tany[:, None], bgv],
                     dim=1).contiguous()


def look_at_w2c(eye: Sequence[float], target: Sequence[float], up=(0.0, -1.0, 0.0)) -> torch.Tensor:
    """OpenCV-style (x right, y down, z forward) world->camera matrix."""
    e = torch.tensor(eye, dtype=torch.float64)
    t = torch.tensor(target, dtype=torch.float64)
    u = torch.tensor(up, dtype=torch.float64)
    z = (t - e) / (t - e).norm()
    x = torch.linalg.cross(-u, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z], dim=0)
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ e
    return w2c.float()


def intrinsics(f: float, cx: float, cy: float, skew: float = 0.0) -> torch.Tensor:
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1] = f, f, cx, cy, skew
    return K
