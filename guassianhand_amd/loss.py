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


class _FitImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, alpha, gt_rgb, gt_mask, bbox_mask, lambda_l1, lambda_mloss, scale):
        if not image.is_cuda:
            raise RuntimeError("gh_fit_loss runs on a ROCm device only (there is no CPU path)")
        L = _lib.lib()
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        im, al, gr, gm, bb = f32(image), f32(alpha), f32(gt_rgb), f32(gt_mask), f32(bbox_mask)
        NV, _, H, W = im.shape
        if al.shape != (NV, H, W) or gr.shape != (NV, H, W, 3) or gm.shape != (NV, H, W) or (bb is not None and bb.shape != (NV, H, W)):
            raise ValueError("fit_image_loss: image (Nv,3,H,W), alpha (Nv,H,W), gt_rgb (Nv,H,W,3), gt_mask / bbox (Nv,H,W)")
        loss = torch.empty((), dtype=torch.float32, device=im.device)
        dimg, dal = torch.empty_like(im), torch.empty_like(al)
        nblk = max(1, min(_N_PARTIALS, (NV * H * W + 255) // 256))
        partials = torch.empty(nblk, dtype=torch.float32, device=im.device)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(im.device):
            rc = L.gh_fit_loss(p(im), p(al), p(gr), p(gm), p(bb), NV, H, W, float(lambda_l1), float(lambda_mloss), float(scale),
                               p(loss), p(dimg), p(dal), p(partials), nblk, C.c_void_p(torch.cuda.current_stream(im.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gh_fit_loss failed: {_abi.status_name(rc)}")
        ctx.save_for_backward(dimg, dal)
        return loss

    @staticmethod
    def backward(ctx, g):
        dimg, dal = ctx.saved_tensors
        return dimg * g, dal * g, None, None, None, None, None, None


def fit_image_loss(image, alpha, gt_rgb, gt_mask, bbox_mask=None, lambda_l1: float = 10.0, lambda_mloss: float = 1.0,
                   scale: float = 1.0) -> torch.Tensor:
    """scale * sum over views of [lambda_l1 * L1(rgb, gt) + lambda_mloss * MSE(clip(alpha, -0.001, 1), gt_mask)] — the
    image part of the reference's fit loss (fit.fit_loss is the torch restatement) — on the rasteriser's own layouts:
    image (Nv,3,H,W), alpha (Nv,H,W); gt_rgb (Nv,H,W,3), gt_mask / bbox_mask (Nv,H,W). Differentiable w.r.t. image, alpha."""
    return _FitImageLoss.apply(image, alpha, gt_rgb, gt_mask, bbox_mask, lambda_l1, lambda_mloss, scale)
