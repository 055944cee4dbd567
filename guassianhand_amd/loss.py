"""Fused image loss (gh_l1_loss, include/gh_raster.h): L = mean|img - gt| with dL/dimg produced by the same pass.

Counterpart of the L1 term of the reference's loss (utils.py:282-294, `lambda_l1 * l1_loss(rgb, gt)`), as used by
bench.py's metric definition. ROCm tensors only; raises if the HIP library is missing."""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi, _lib

_N_PARTIALS = 1024


def _l1_kernel(image, target, guard=None):
    """(mean|image - target|, its gradient w.r.t. image) from one pass (gh_l1_loss). guard: device GhCounters of the
    forward that rendered `image` (overflow -> loss NaN, zero gradient)."""
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
                          None if guard is None else C.c_void_p(guard.data_ptr()),
                          C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gh_l1_loss failed: {_abi.status_name(rc)}")
    return loss, grad


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target):
        loss, grad = _l1_kernel(image, target)
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


def _fit_kernel(image, alpha, gt_rgb, gt_mask, bbox_mask, lambda_l1, lambda_mloss, scale, guard=None):
    """(loss, dL/dimage, dL/dalpha) of the fit's image loss from one pass (gh_fit_loss)."""
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
                           p(loss), p(dimg), p(dal), p(partials), nblk, p(guard),
                           C.c_void_p(torch.cuda.current_stream(im.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gh_fit_loss failed: {_abi.status_name(rc)}")
    return loss, dimg, dal


class _FitImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, alpha, gt_rgb, gt_mask, bbox_mask, lambda_l1, lambda_mloss, scale, guard):
        loss, dimg, dal = _fit_kernel(image, alpha, gt_rgb, gt_mask, bbox_mask, lambda_l1, lambda_mloss, scale, guard)
        ctx.save_for_backward(dimg, dal)
        return loss

    @staticmethod
    def backward(ctx, g):
        dimg, dal = ctx.saved_tensors
        return dimg * g, dal * g, None, None, None, None, None, None, None


def fit_image_loss(image, alpha, gt_rgb, gt_mask, bbox_mask=None, lambda_l1: float = 10.0, lambda_mloss: float = 1.0,
                   scale: float = 1.0, guard=None) -> torch.Tensor:
    """scale * sum over views of [lambda_l1 * L1(rgb, gt) + lambda_mloss * MSE(clip(alpha, -0.001, 1), gt_mask)] — the
    image part of the reference's fit loss (fit.fit_loss is the torch restatement) — on the rasteriser's own layouts:
    image (Nv,3,H,W), alpha (Nv,H,W); gt_rgb (Nv,H,W,3), gt_mask / bbox_mask (Nv,H,W). Differentiable w.r.t. image, alpha."""
    return _FitImageLoss.apply(image, alpha, gt_rgb, gt_mask, bbox_mask, lambda_l1, lambda_mloss, scale, guard)


# ---------------------------------------------------------------------------------------------------
class _RenderedLoss(torch.autograd.Function):
    """Render + image loss as ONE autograd node. The loss kernel leaves dL/dimage (and dL/dalpha) unscaled; the upstream
    dL/dloss reaches the render backward as a device scalar (GhGrads.upstream_scale) and is applied while the kernel reads
    its pixels, so the autograd product `dL/dimage * dL/dloss` costs no pass over the images."""

    @staticmethod
    def forward(ctx, spec, cache, cams, H, W, sh_degree, scale_modifier, use_rgb, sync, max_instances, per_view, defer_loss, xyz, opacity,
                scaling, rotation, shs, xyz_b, opacity_b, color_w, color_b):
        from . import rasterizer as R
        kind = spec[0]
        kw = dict(colors_precomp=shs.reshape(shs.shape[0], 3)) if use_rgb else dict(shs=shs)
        if kind == "l1":                             # the fused epilogue reads the target as it lies: float32, contiguous, image-shaped
            tgt = spec[1].detach().float().contiguous()
            if tuple(tgt.shape) != (cams.reshape(-1, 40).shape[0], 3, H, W):
                raise ValueError("image and target must have the same shape")
            spec = ("l1", tgt)
        elif kind == "fit":
            f32 = lambda t: None if t is None else t.detach().float().contiguous()
            spec = ("fit", f32(spec[1]), f32(spec[2]), f32(spec[3])) + tuple(spec[4:])
        image, radii, rctx = R.cached_raster_forward(cache, cams, xyz, opacity, scaling, rotation, H=H, W=W, sh_degree=sh_degree,
                                                     scale_modifier=scale_modifier, xyz_b=xyz_b, opacity_b=opacity_b,
                                                     color_w=color_w, color_b=color_b, sync=sync, max_instances=max_instances,
                                                     return_alpha=(kind == "fit"), per_view_gaussians=per_view,
                                                     l1_target=spec[1] if kind == "l1" else None,
                                                     fit_loss=spec[1:] if kind == "fit" else None, defer_loss=defer_loss, **kw)
        guard = rctx.ws[:16]                          # device-side overflow guard: an overflowed render yields loss NaN, zero gradients
        if kind == "l1":
            # the render kernel's epilogue has produced both (GhOutputs.l1_*) wherever the library fuses them; else one pass over the image
            loss, dimg = rctx.l1 if rctx.l1 is not None else _l1_kernel(image, spec[1], guard)
            dal = None
        elif kind == "fit":
            loss, dimg, dal = rctx.fit if rctx.fit is not None else _fit_kernel(image, rctx.alpha, *spec[1:], guard=guard)
        else:
            raise ValueError(kind)
        ctx.rctx, ctx.dimg, ctx.dal, ctx.use_rgb = rctx, dimg, dal, use_rgb
        # (the backward kernels re-read the inputs through the context's pointers: autograd's version check, as in rasterizer.py)
        ctx.save_for_backward(*[t for t in (cams, xyz, opacity, scaling, rotation, shs, xyz_b, opacity_b, color_w, color_b) if t is not None])
        ctx.set_materialize_grads(False)             # no image-sized zero tensors for the outputs that carry no gradient
        ctx.shapes = [None if t is None else t.shape for t in (xyz, opacity, scaling, rotation, shs, xyz_b, opacity_b, color_w, color_b)]
        alpha = rctx.alpha if rctx.alpha is not None else image.new_zeros(0)
        ctx.mark_non_differentiable(image, alpha, radii)
        return loss, image, alpha, radii

    @staticmethod
    def backward(ctx, g_loss, _gi, _ga, _gr):
        from . import rasterizer as R
        ctx.saved_tensors
        g = R.raster_backward(ctx.rctx, ctx.dimg, want_means2D=False, dL_dalpha=ctx.dal, grad_scale=g_loss,
                              want=R._wanted(ctx.needs_input_grad[12:21], ctx.use_rgb))
        ctx.rctx = None
        s = ctx.shapes
        col = g.get("colors_precomp" if ctx.use_rgb else "shs")
        opt = lambda k, i: g[k].reshape(s[i]) if (s[i] is not None and k in g) else None
        return (None,) * 12 + (opt("means3D", 0), opt("opacities", 1), opt("scales", 2), opt("rotations", 3),
                              None if col is None else col.reshape(s[4]), opt("xyz_b", 5), opt("opacity_b", 6),
                              opt("color_w", 7), opt("color_b", 8))


def _rendered_loss(spec, cams, xyz, opacity, scaling, rotation, shs, *, H, W, use_rgb, sh_degree=3, scale_modifier=1.0, xyz_b=None,
                   opacity_b=None, color_w=None, color_b=None, sync=True, max_instances=None, per_view_gaussians=False,
                   geometry_cache=None, depth_bound=None, defer_loss=False):
    if depth_bound is not None:                        # rasterizer.DepthBoundCache: moving geometry (see rasterize_views)
        if geometry_cache is not None:
            raise ValueError("geometry_cache (static geometry) or depth_bound (moving geometry), not both")
        geometry_cache = depth_bound
    return _RenderedLoss.apply(spec, geometry_cache, cams, int(H), int(W), int(sh_degree if not use_rgb else 0), float(scale_modifier), bool(use_rgb),
                               bool(sync), max_instances, bool(per_view_gaussians), bool(defer_loss), xyz, opacity, scaling, rotation, shs, xyz_b,
                               opacity_b, color_w, color_b)


def rendered_l1_loss(cams, xyz, opacity, scaling, rotation, shs, target, **kw):
    """mean|render - target| with the render (rasterizer.rasterize_views arguments) and the loss in one autograd node.
    Returns (loss, image (Nv,3,H,W) detached, radii). Same values as l1_mean_loss(rasterize_views(...)[0], target).
    defer_loss=True (GH_FLAG_DEFER_LOSS_SUM): where the loss comes from the render kernel's epilogue its final one-workgroup sum
    moves into the BACKWARD (a spare workgroup of the render backward): the returned loss tensor holds its value only once
    loss.backward() has been enqueued — for loops that look at the loss after the step (bench.py, the fit's log)."""
    loss, image, _alpha, radii = _rendered_loss(("l1", target), cams, xyz, opacity, scaling, rotation, shs, **kw)
    return loss, image, radii


def rendered_fit_loss(cams, xyz, opacity, scaling, rotation, shs, gt_rgb, gt_mask, bbox_mask=None, lambda_l1: float = 10.0,
                      lambda_mloss: float = 1.0, scale: float = 1.0, **kw):
    """The fit's image loss (fit_image_loss) of a fused RGB + alpha render, one autograd node.
    Returns (loss, image (Nv,3,H,W), alpha (Nv,H,W)) with image / alpha detached."""
    loss, image, alpha, _radii = _rendered_loss(("fit", gt_rgb, gt_mask, bbox_mask, float(lambda_l1), float(lambda_mloss), float(scale)),
                                                cams, xyz, opacity, scaling, rotation, shs, **kw)
    return loss, image, alpha
