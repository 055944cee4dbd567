"""Per-Gaussian bilinear lookup of the learnable UV maps.

Host side of gh_uv_sample_forward / gh_uv_sample_backward (include/gh_raster.h): the device counterpart of
`query_triplane_texture` (tgs/models/renderer_one_shot.py:420-446, F.grid_sample bilinear, align_corners=True) at the
call sites :489-492. Maps are kept CHANNEL-LAST (Hm, Wm, C); `to_channel_last` / `to_reference_layout` convert to and
from the reference's (C, Hm, Wm) parameters (infer_one_shot.py:160,163).
On ROCm tensors the HIP kernels run (and raise if the library is missing); CPU tensors — used only by the CPU tests of
the fit-loop host logic — go through torch's grid_sample, which the GPU test compares the kernels against.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _abi, _lib


def to_channel_last(param_chw: torch.Tensor) -> torch.Tensor:
    return param_chw.permute(1, 2, 0).contiguous()


def to_reference_layout(map_hwc: torch.Tensor) -> torch.Tensor:
    return map_hwc.permute(2, 0, 1)


class _UvSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, map_hwc, uv):
        L = _lib.lib()
        m = map_hwc.detach().float().contiguous()
        u = uv.detach().float().contiguous()
        Hm, Wm, Cc = m.shape
        P = u.shape[0]
        out = torch.empty(P, Cc, dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            rc = L.gh_uv_sample_forward(C.c_void_p(m.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(out.data_ptr()), P, Cc,
                                        Hm, Wm, C.c_void_p(torch.cuda.current_stream(m.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gh_uv_sample_forward failed: {_abi.status_name(rc)}")
        ctx.save_for_backward(u)
        ctx.shape = (Hm, Wm, Cc)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.lib()
        (u,) = ctx.saved_tensors
        Hm, Wm, Cc = ctx.shape
        g = grad_out.detach().float().contiguous()
        dmap = torch.zeros(Hm, Wm, Cc, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = L.gh_uv_sample_backward(C.c_void_p(u.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(dmap.data_ptr()),
                                         u.shape[0], Cc, Hm, Wm, C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gh_uv_sample_backward failed: {_abi.status_name(rc)}")
        return dmap, None


def uv_sample(map_hwc: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    """(Hm,Wm,C) map sampled at uv (P,2) in [-1,1] -> (P,C); differentiable w.r.t. the map."""
    if map_hwc.is_cuda:
        return _UvSample.apply(map_hwc, uv)
    out = F.grid_sample(map_hwc.permute(2, 0, 1)[None], uv[None, :, None, :], align_corners=True, mode="bilinear")
    return out[0, :, :, 0].transpose(0, 1)
