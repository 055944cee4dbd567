"""Import-name shim: lets the reference's line `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer` (tgs/models/renderer_one_shot.py:3) resolve to the
MI355X-native rasteriser without touching the reference source. See INTEGRATION.md."""
from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
