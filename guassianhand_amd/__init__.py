"""guassianhand_amd — MI355X-native differentiable Gaussian-splatting rasteriser for GaussianHand.

Drop-in for the `diff_gaussian_rasterization` import of tgs/models/renderer_one_shot.py:3 (the hot path
named by BASELINE.json `north_star`). The compute lives in hand-written HIP kernels for gfx950 behind the
C-ABI of include/gh_raster.h; this package is the thin PyTorch-ROCm host side.
"""
__version__ = "0.1.0"
