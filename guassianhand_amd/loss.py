"""Fused image loss (gh_l1_loss, include/gh_raster.h): L = mean|img - gt| with dL/dimg produced by the same pass.

Counterpart of the L1 term of the reference's loss (utils.py:282-294, `lambda_l1 * l1_loss(rgb, gt)`), as used by
bench.py's metric definition. ROCm tensors only; raises if the HIP library is missing."""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi, _lib

_N_PARTIALS = 1024


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target):
        if not image.is_cuda:
            raise RuntimeError("gh_l1_loss runs on a ROCm device only (there is no CPU path)")
        L = _lib.lib()
        a = image.detach().float().contiguous()
        b = target.detach().float().contiguous()
        if a.shape != b.shape:
            raise ValueError("image and target must have the same shape")
        n = a.numel()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a)
        nblk = max(1, min(_N_PARTIALS, (n // 4 + 255) // 256))
        partials = torch.empty(nblk, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            rc = L.gh_l1_loss(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, C.c_void_p(loss.data_ptr()),
                              C.c_void_p(grad.data_ptr()), C.c_void_p(partials.data_ptr()), nblk,
                              C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gh_l1_loss failed: {_abi.status_name(rc)}")
        ctx.save_for_backward(grad)
        ctx.shape = image.shape
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        return (grad * grad_loss).reshape(ctx.shape), None


def l1_mean_loss(image: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean|image - target| (0-dim tensor), differentiable w.r.t. image."""
    return _L1Mean.apply(image, target)
