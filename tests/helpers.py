"""Shared helpers for the parity tests (oracle = checker, HIP path = thing under test)."""
import torch

from guassianhand_amd.scenes import make_scene


def scene_kwargs(sc, on_cpu=True):
    """(colour kwargs, blend kwargs) for OracleRender / raster_forward from a Scene."""
    kw = dict(colors_precomp=sc.shs.squeeze(1)) if sc.use_rgb else dict(shs=sc.shs, sh_degree=sc.sh_degree)
    bl = {k: getattr(sc, k) for k in ("xyz_b", "opacity_b", "color_w", "color_b") if getattr(sc, k) is not None}
    return kw, bl


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b, floor=1e-3):
    """max |a-b| / (|b| + floor*max|b|): the 'grad rtol' of BASELINE.json with a scale-aware floor."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(((a - b).abs() / (b.abs() + floor * b.abs().max() + 1e-30)).max())


def dimg_like(nv, H, W, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(nv, 3, H, W, generator=g)
