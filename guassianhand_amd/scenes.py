"""Seeded synthetic workloads of SURVEY.md §8(d) / BASELINE.md §3.

No MANO weights or InterHand2.6M data exist here, so the hands are a capsule-union surrogate with the
reference's Gaussian counts (49,281 per hand = 778-vertex MANO template subdivided 3x:
dataset_one_shot.py:321-325, mis_utils.py:45-122; two hands 98,562: dataset_one_shot.py:371,404-408).
Everything is generated on the CPU from one torch.Generator so every rank/device sees the same bytes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

from .camera import intrinsics, look_at_w2c, pack_cameras_from_w2c

SEED = 20240610
P_HAND = 49281


@dataclass
class Scene:
    xyz: torch.Tensor          # (P,3)
    opacity: torch.Tensor      # (P,1)  post-sigmoid
    rotation: torch.Tensor     # (P,4)  unit quaternion (w,x,y,z)
    scaling: torch.Tensor      # (P,3)  post-exp
    shs: torch.Tensor          # (P,1,3) RGB (use_rgb) or (P,16,3)
    color_w: Optional[torch.Tensor]   # (48,)
    color_b: Optional[torch.Tensor]   # (P,48)
    opacity_b: Optional[torch.Tensor] # (P,1)
    xyz_b: Optional[torch.Tensor]     # (3,)
    w2c: torch.Tensor          # (Nv,4,4)
    K: torch.Tensor            # (Nv,4,4)
    H: int
    W: int
    bg: torch.Tensor           # (3,)
    use_rgb: bool
    sh_degree: int

    def to(self, device) -> "Scene":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return Scene(**kw)

    @property
    def P(self) -> int:
        return self.xyz.shape[0]

    def cams(self) -> torch.Tensor:
        return pack_cameras_from_w2c(self.w2c, self.K, self.H, self.W, self.bg)


def _capsule_points(n: int, a: torch.Tensor, b: torch.Tensor, r: float, g: torch.Generator) -> torch.Tensor:
    """n points on the surface of the capsule with axis a->b, radius r (cylinder part + end caps)."""
    axis = b - a
    L = float(axis.norm())
    ez = axis / L
    tmp = torch.tensor([1.0, 0.0, 0.0]) if abs(float(ez[0])) < 0.9 else torch.tensor([0.0, 1.0, 0.0])
    ex = torch.linalg.cross(ez, tmp)
    ex = ex / ex.norm()
    ey = torch.linalg.cross(ez, ex)
    area_cyl, area_caps = 2 * math.pi * r * L, 4 * math.pi * r * r
    n_cyl = int(round(n * area_cyl / (area_cyl + area_caps)))
    n_cap = n - n_cyl
    th = torch.rand(n_cyl, generator=g) * 2 * math.pi
    h = torch.rand(n_cyl, generator=g) * L
    cyl = a + h[:, None] * ez + r * (torch.cos(th)[:, None] * ex + torch.sin(th)[:, None] * ey)
    v = torch.randn(n_cap, 3, generator=g)
    v = v / v.norm(dim=1, keepdim=True)
    along = v @ ez
    cap = torch.where(along[:, None] > 0, b + r * v, a + r * v)
    return torch.cat([cyl, cap], dim=0)


def _box_points(n: int, half: torch.Tensor, g: torch.Generator) -> torch.Tensor:
    """n points on the surface of an axis-aligned box of half-extents `half`, centred at 0."""
    areas = torch.tensor([half[1] * half[2], half[0] * half[2], half[0] * half[1]]) * 4
    probs = torch.cat([areas, areas]) / (2 * areas.sum())
    face = torch.multinomial(probs, n, replacement=True, generator=g)
    u = (torch.rand(n, 3, generator=g) * 2 - 1) * half
    ax = face % 3
    sign = torch.where(face < 3, 1.0, -1.0)
    u[torch.arange(n), ax] = sign * half[ax]
    return u


def hand_surrogate(n: int, g: torch.Generator, curl: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Right-hand surrogate: palm box 9x9x2.5 cm + 5 fingers x 3 phalanges (capsules r 0.7-1.0 cm).
    curl: (5,) finger-curl angles in rad (config-5 'mixed poses'); None = straight fingers."""
    parts = []
    palm_half = torch.tensor([0.045, 0.045, 0.0125])
    finger_x = [-0.036, -0.018, 0.0, 0.018, 0.040]
    lengths = [[0.040, 0.025, 0.020], [0.045, 0.028, 0.022], [0.042, 0.026, 0.021], [0.034, 0.021, 0.019],
               [0.038, 0.030, 0.026]]
    radii = [[0.0085, 0.0078, 0.0070], [0.0090, 0.0082, 0.0072], [0.0088, 0.0080, 0.0071],
             [0.0080, 0.0074, 0.0070], [0.0100, 0.0092, 0.0085]]
    # area-proportional point budget
    areas = [float(8 * (palm_half[0] * palm_half[1] + palm_half[0] * palm_half[2] + palm_half[1] * palm_half[2]))]
    for f in range(5):
        for j in range(3):
            areas.append(2 * math.pi * radii[f][j] * lengths[f][j] + 4 * math.pi * radii[f][j] ** 2)
    tot = sum(areas)
    counts = [int(n * a / tot) for a in areas]
    counts[0] += n - sum(counts)
    parts.append(_box_points(counts[0], palm_half, g))
    k = 1
    for f in range(5):
        base = torch.tensor([finger_x[f], 0.045, 0.0])
        direction = torch.tensor([0.0, 1.0, 0.0])
        if f == 4:  # thumb sticks out sideways
            base = torch.tensor([0.045, -0.015, 0.0])
            direction = torch.tensor([0.8, 0.6, 0.0])
            direction = direction / direction.norm()
        ang = 0.0
        for j in range(3):
            if curl is not None:
                ang += float(curl[f]) * (0.5 if j == 0 else 1.0)
            # curl rotates the phalanx about the x axis, bending towards -z
            d = torch.tensor([direction[0], direction[1] * math.cos(ang), -abs(direction[1]) * math.sin(ang)])
            d = d / d.norm()
            tip = base + lengths[f][j] * d
            parts.append(_capsule_points(counts[k], base, tip, radii[f][j], g))
            base = tip
            k += 1
    return torch.cat(parts, dim=0)


def _two_hands(n_per_hand: int, g: torch.Generator, curl_r=None, curl_l=None) -> torch.Tensor:
    right = hand_surrogate(n_per_hand, g, curl_r)
    left = hand_surrogate(n_per_hand, g, curl_l)
    left = left * torch.tensor([-1.0, 1.0, 1.0])                       # mirror
    a = math.radians(30.0)
    Rz = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    left = left @ Rz.T + torch.tensor([-0.06, 0.0, 0.02])              # offset + rotate: fingers interleave
    return torch.cat([right, left], dim=0)                             # right-then-left (dataset_one_shot.py:404-408)


def ring_cameras(center: torch.Tensor, n_views: int, H: int, W: int, f: float, radius: float = 1.0):
    """8 novel views on a ring: elevations {0,20} deg x azimuths {0,90,180,270} deg."""
    w2cs, Ks = [], []
    for v in range(n_views):
        az = math.radians(90.0 * (v % 4) + 11.25 * (v // 8))   # views beyond the 8 of §8d: same ring, rotated
        el = math.radians(20.0 * ((v // 4) % 2))
        eye = center + radius * torch.tensor([math.sin(az) * math.cos(el), -math.sin(el), -math.cos(az) * math.cos(el)])
        w2cs.append(look_at_w2c(eye.tolist(), center.tolist()))
        Ks.append(intrinsics(f, W / 2.0, H / 2.0))
    return torch.stack(w2cs), torch.stack(Ks)


def _attributes(P: int, g: torch.Generator, use_rgb: bool, scale_mean: float, blend: bool):
    scaling = torch.exp(scale_mean + 0.35 * torch.randn(P, 3, generator=g))
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opacity = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g))
    if use_rgb:
        shs = torch.rand(P, 1, 3, generator=g)
    else:
        shs = 0.3 * torch.randn(P, 16, 3, generator=g)
        shs[:, 0, :] += 0.5
    if blend:
        color_w = 1.0 + 0.05 * torch.randn(48, generator=g)
        color_b = 0.02 * torch.randn(P, 48, generator=g)
        opacity_b = 0.02 * torch.randn(P, 1, generator=g)
        xyz_b = torch.zeros(3)
    else:
        color_w = color_b = opacity_b = xyz_b = None
    return scaling, rot, opacity, shs, color_w, color_b, opacity_b, xyz_b


def make_scene(config: str, n_views: int = 1, seed: int = SEED, scale_mean: float = -6.2,
               blend: Optional[bool] = None, P: Optional[int] = None, use_rgb: Optional[bool] = None) -> Scene:
    """config: 'random1k' (BASELINE configs[0]), 'one_hand' ([1]), 'two_hands' ([2]/[3]),
    'two_hands_hd' ([4]: 1024x1024, SH degree 3, mixed poses)."""
    g = torch.Generator().manual_seed(seed)
    if config == "random1k":
        P = P or 1000
        xyz = torch.rand(P, 3, generator=g) * 0.2 - 0.1
        H = W = 128
        f = 325.0
        rgb = True if use_rgb is None else use_rgb
        deg = 0 if rgb else 3
        blend = False if blend is None else blend
        centre = torch.zeros(3)
        scale_mean = -5.0 if scale_mean == -6.2 else scale_mean
    elif config == "one_hand":
        xyz = hand_surrogate(P or P_HAND, g)
        H, W, f = 512, 334, 1300.0
        rgb = True if use_rgb is None else use_rgb
        deg = 0 if rgb else 3
        blend = False if blend is None else blend
        centre = xyz.mean(0)
    elif config == "two_hands":
        xyz = _two_hands((P or 2 * P_HAND) // 2, g)
        H, W, f = 512, 334, 1300.0
        rgb = True if use_rgb is None else use_rgb
        deg = 0 if rgb else 3
        blend = True if blend is None else blend
        centre = xyz.mean(0)
    elif config == "two_hands_hd":
        curl_r = torch.rand(5, generator=g) * 1.2
        curl_l = torch.rand(5, generator=g) * 1.2
        xyz = _two_hands((P or 2 * P_HAND) // 2, g, curl_r, curl_l)
        H, W, f = 1024, 1024, 2600.0
        rgb = False if use_rgb is None else use_rgb
        deg = 0 if rgb else 3
        blend = True if blend is None else blend
        centre = xyz.mean(0)
    else:
        raise ValueError(config)
    Pn = xyz.shape[0]
    scaling, rot, opacity, shs, color_w, color_b, opacity_b, xyz_b = _attributes(Pn, g, rgb, scale_mean, blend)
    w2c, K = ring_cameras(centre, n_views, H, W, f)
    return Scene(xyz=xyz.float(), opacity=opacity, rotation=rot, scaling=scaling, shs=shs, color_w=color_w,
                 color_b=color_b, opacity_b=opacity_b, xyz_b=xyz_b, w2c=w2c, K=K, H=H, W=W, bg=torch.zeros(3),
                 use_rgb=rgb, sh_degree=deg)


def perturbed_target_xyz(scene: Scene, sigma: float = 1e-3, seed: int = SEED + 1) -> torch.Tensor:
    """Positions +N(0, 1 mm): the GT image is a render of this copy so gradients are non-trivial (§8d)."""
    g = torch.Generator().manual_seed(seed)
    return scene.xyz.cpu() + sigma * torch.randn(scene.xyz.shape, generator=g)
