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
from typing import Optional

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
        # Deterministic: the texels under the footprints are listed once per UV tensor (ActiveTexels, cached on identity +
        # version) and the scatter runs as a fixed-order gather (gh_uv_scatter_sorted) — no float atomics on the fit path.
        (u,) = ctx.saved_tensors
        Hm, Wm, Cc = ctx.shape
        at = _texels_of(u, Hm, Wm)
        g = grad_out.detach().float().contiguous()
        dtex = torch.zeros(at.U, Cc, dtype=torch.float32, device=g.device)
        uv_gather_backward(g, at, dtex)
        return at.dense(dtex), None


_texel_cache = []      # [(weakref to the uv tensor, version, Hm, Wm, ActiveTexels)], most recent first


def _texels_of(uv: torch.Tensor, Hm: int, Wm: int) -> "ActiveTexels":
    import weakref
    for i, (ref, ver, h, w, at) in enumerate(_texel_cache):
        if ref() is uv and ver == uv._version and (h, w) == (Hm, Wm):
            return at
    at = ActiveTexels(uv, Hm, Wm)
    _texel_cache.insert(0, (weakref.ref(uv), uv._version, Hm, Wm, at))
    del _texel_cache[4:]
    return at


def uv_sample(map_hwc: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    """(Hm,Wm,C) map sampled at uv (P,2) in [-1,1] -> (P,C); differentiable w.r.t. the map."""
    if map_hwc.is_cuda:
        return _UvSample.apply(map_hwc, uv)
    out = F.grid_sample(map_hwc.permute(2, 0, 1)[None], uv[None, :, None, :], align_corners=True, mode="bilinear")
    return out[0, :, :, 0].transpose(0, 1)


# ---- active-texel form (gh_uv_gather_* / gh_adam_reg_step) ------------------------------------------------------------
class ActiveTexels:
    """The texels of an (Hm, Wm) map that lie under the bilinear footprints of fixed UV coordinates.

    During a one-shot fit the UVs never change, so these U <= 4P texels are the only ones that ever receive an image
    gradient; all other texels of the zero-initialised maps keep gradient 0 under the regularisers
    (infer_one_shot.py:514-518) and stay exactly 0 under Adam. `index` (U,) int64 = linear texel index y*Wm + x
    (sorted), `slot` (P,4) int32 = compact row of the nw/ne/sw/se corner or -1 outside the map, `w` (P,4) the bilinear
    weights — the same float32 arithmetic as gh_bilinear in csrc/gh_uv.hip, so a gather over compacted texels is
    bit-identical to gh_uv_sample_forward over the dense map."""

    def __init__(self, uv: torch.Tensor, Hm: int, Wm: int):
        uv = uv.detach().float()
        ix = ((uv[:, 0] + 1.0) * 0.5) * float(Wm - 1)
        iy = ((uv[:, 1] + 1.0) * 0.5) * float(Hm - 1)
        fx, fy = torch.floor(ix), torch.floor(iy)
        x0, y0 = fx.long(), fy.long()
        wx1, wy1 = ix - fx, iy - fy
        wx0, wy0 = 1.0 - wx1, 1.0 - wy1
        xs = torch.stack([x0, x0 + 1, x0, x0 + 1], 1)
        ys = torch.stack([y0, y0, y0 + 1, y0 + 1], 1)
        valid = (xs >= 0) & (xs < Wm) & (ys >= 0) & (ys < Hm)
        lin = ys * Wm + xs
        self.index = torch.unique(lin[valid])
        slot = torch.searchsorted(self.index, lin.clamp(0, Hm * Wm - 1))
        self.slot = torch.where(valid, slot, torch.full_like(slot, -1)).to(torch.int32).contiguous()
        self.w = torch.stack([wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1], 1).contiguous()
        self.Hm, self.Wm, self.P = Hm, Wm, uv.shape[0]
        # the same incidence transposed (CSR by texel) for the deterministic backward (gh_uv_scatter_sorted): pairs =
        # 4 * gaussian + corner of every valid corner, grouped by texel, ascending inside a texel
        flat = self.slot.reshape(-1).long()
        pairs = torch.nonzero(flat >= 0).reshape(-1)
        order = torch.sort(flat[pairs], stable=True).indices
        self.pairs = pairs[order].to(torch.int32).contiguous()
        counts = torch.bincount(flat[pairs], minlength=int(self.index.numel()))
        self.row_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32).contiguous()

    @property
    def U(self) -> int:
        return int(self.index.numel())

    def compact(self, map_hwc: torch.Tensor) -> torch.Tensor:
        """(Hm,Wm,C) dense map -> (U,C) active texels."""
        return map_hwc.reshape(self.Hm * self.Wm, -1)[self.index].contiguous()

    def dense(self, texels: torch.Tensor) -> torch.Tensor:
        """(U,C) active texels -> (Hm,Wm,C) dense map (zeros elsewhere)."""
        out = torch.zeros(self.Hm * self.Wm, texels.shape[1], dtype=texels.dtype, device=texels.device)
        out[self.index] = texels
        return out.view(self.Hm, self.Wm, -1)


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_device(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("the active-texel kernels run on a ROCm device only (there is no CPU path)")
        if t.dtype not in (torch.float32, torch.int32) or not t.is_contiguous():
            raise ValueError("active-texel kernels take contiguous float32 / int32 tensors")


def uv_gather(texels: torch.Tensor, at: ActiveTexels) -> torch.Tensor:
    """(U,C) active texels -> per-Gaussian values (P,C); no autograd (the fit loop calls uv_gather_backward itself)."""
    _need_device(texels, at.slot, at.w)
    L = _lib.lib()
    Cc = texels.shape[1]
    out = torch.empty(at.P, Cc, dtype=torch.float32, device=texels.device)
    with torch.cuda.device(texels.device):
        rc = L.gh_uv_gather_forward(C.c_void_p(texels.data_ptr()), C.c_void_p(at.slot.data_ptr()), C.c_void_p(at.w.data_ptr()),
                                    C.c_void_p(out.data_ptr()), at.P, Cc, _stream(texels))
    if rc != 0:
        raise RuntimeError(f"gh_uv_gather_forward failed: {_abi.status_name(rc)}")
    return out


def uv_gather_backward(grad_out: torch.Tensor, at: ActiveTexels, grad_texels: torch.Tensor) -> None:
    """Accumulates d(loss)/d(texels) (U,C) from d(loss)/d(per-Gaussian values) (P,C): a fixed-order gather over each
    texel's (Gaussian, corner) list (gh_uv_scatter_sorted) — bitwise reproducible, no atomics."""
    g = grad_out.detach().float().contiguous()
    _need_device(g, grad_texels, at.row_ptr, at.pairs, at.w)
    assert g.shape == (at.P, grad_texels.shape[1]) and grad_texels.shape[0] == at.U
    L = _lib.lib()
    with torch.cuda.device(g.device):
        rc = L.gh_uv_scatter_sorted(C.c_void_p(at.row_ptr.data_ptr()), C.c_void_p(at.pairs.data_ptr()), C.c_void_p(at.w.data_ptr()),
                                    C.c_void_p(g.data_ptr()), C.c_void_p(grad_texels.data_ptr()), at.U, g.shape[1], _stream(g))
    if rc != 0:
        raise RuntimeError(f"gh_uv_scatter_sorted failed: {_abi.status_name(rc)}")


def uv_gather2(texels_a: torch.Tensor, texels_b: torch.Tensor, at: ActiveTexels):
    """uv_gather of two maps that share the texel index, in one launch (gh_uv_gather_forward2): ((P,Ca), (P,Cb))."""
    _need_device(texels_a, texels_b, at.slot, at.w)
    L = _lib.lib()
    Ca, Cb = texels_a.shape[1], texels_b.shape[1]
    oa = torch.empty(at.P, Ca, dtype=torch.float32, device=texels_a.device)
    ob = torch.empty(at.P, Cb, dtype=torch.float32, device=texels_a.device)
    with torch.cuda.device(texels_a.device):
        rc = L.gh_uv_gather_forward2(C.c_void_p(texels_a.data_ptr()), Ca, C.c_void_p(texels_b.data_ptr()), Cb, C.c_void_p(at.slot.data_ptr()),
                                     C.c_void_p(at.w.data_ptr()), C.c_void_p(oa.data_ptr()), C.c_void_p(ob.data_ptr()), at.P, _stream(texels_a))
    if rc != 0:
        raise RuntimeError(f"gh_uv_gather_forward2 failed: {_abi.status_name(rc)}")
    return oa, ob


def uv_gather_backward2(grad_a: torch.Tensor, grad_b: torch.Tensor, at: ActiveTexels, grad_texels_a: torch.Tensor,
                        grad_texels_b: torch.Tensor) -> None:
    """uv_gather_backward of both maps in one launch (gh_uv_scatter_sorted2), bit-identical to two calls."""
    ga, gb = grad_a.detach().float().contiguous(), grad_b.detach().float().contiguous()
    if gb.dim() == 1:
        gb = gb.reshape(-1, 1)
    _need_device(ga, gb, grad_texels_a, grad_texels_b, at.row_ptr, at.pairs, at.w)
    assert ga.shape == (at.P, grad_texels_a.shape[1]) and gb.shape == (at.P, grad_texels_b.shape[1])
    assert grad_texels_a.shape[0] == at.U and grad_texels_b.shape[0] == at.U
    L = _lib.lib()
    with torch.cuda.device(ga.device):
        rc = L.gh_uv_scatter_sorted2(C.c_void_p(at.row_ptr.data_ptr()), C.c_void_p(at.pairs.data_ptr()), C.c_void_p(at.w.data_ptr()),
                                     C.c_void_p(ga.data_ptr()), ga.shape[1], C.c_void_p(grad_texels_a.data_ptr()),
                                     C.c_void_p(gb.data_ptr()), gb.shape[1], C.c_void_p(grad_texels_b.data_ptr()), at.U, _stream(ga))
    if rc != 0:
        raise RuntimeError(f"gh_uv_scatter_sorted2 failed: {_abi.status_name(rc)}")


class AdamReg:
    """State of gh_adam_reg_step for one parameter tensor: torch.optim.Adam's update with the gradient of
    reg_l1*sum|p| + reg_l2*sum(p^2) folded in; `step()` returns (sum|p|, sum p^2) of the pre-update values as a
    2-element device tensor (no host sync)."""
    N_PARTIALS = 1024

    def __init__(self, param: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, reg_l1: float = 0.0,
                 reg_l2: float = 0.0):
        _need_device(param)
        self.param, self.lr, self.betas, self.eps, self.reg_l1, self.reg_l2 = param, lr, betas, eps, reg_l1, reg_l2
        self.grad = torch.zeros_like(param)
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.t = 0
        n = param.numel()
        self.n_partials = max(1, min(self.N_PARTIALS, (n + 255) // 256))
        self.partials = torch.zeros(self.n_partials, 2, dtype=torch.float32, device=param.device)
        self.step_state = torch.zeros(2, dtype=torch.int32, device=param.device)   # steps actually applied (device side)

    def step(self, guard: Optional[torch.Tensor] = None, grad: Optional[torch.Tensor] = None, sums: bool = True):
        """guard: device GhCounters of the render whose gradients are in `.grad` (rasterizer.last_guard()): when its
        overflow flag is set the kernel leaves param / moments untouched and does not count the step.
        grad: use this buffer (same shape, fp32, contiguous; it is cleared like `.grad`) instead of `.grad` — a gradient the
        kernels already wrote somewhere needs no copy. sums=False: do not reduce the block partials (see `sums()`)."""
        self.t += 1
        L = _lib.lib()
        p = self.param
        g = self.grad if grad is None else grad
        if grad is not None:
            _need_device(g)
            assert g.numel() == p.numel()
        with torch.cuda.device(p.device):
            rc = L.gh_adam_reg_step(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
                                    C.c_void_p(self.exp_avg_sq.data_ptr()), p.numel(), self.t, self.lr, self.betas[0], self.betas[1],
                                    self.eps, self.reg_l1, self.reg_l2, C.c_void_p(self.partials.data_ptr()), self.n_partials,
                                    None if guard is None else C.c_void_p(guard.data_ptr()),
                                    C.c_void_p(self.step_state.data_ptr()), _stream(p))
        if rc != 0:
            raise RuntimeError(f"gh_adam_reg_step failed: {_abi.status_name(rc)}")
        return self.partials.sum(0) if sums else None

    def sums(self) -> torch.Tensor:
        """(sum|p|, sum p^2) of the values the last step() started from."""
        return self.partials.sum(0)


def adam_group_step(adams, guard: Optional[torch.Tensor] = None, grads=None) -> None:
    """One launch for the step of up to four AdamReg states that share lr / betas / eps (gh_adam_reg_step_group): the same
    updates, moments, step counts and block partials as stepping them one by one. grads[i] (optional) replaces adams[i].grad."""
    L = _lib.lib()
    a0 = adams[0]
    arr = (_abi.GhAdamTensor * len(adams))()
    for i, a in enumerate(adams):
        a.t += 1
        assert (a.lr, a.betas, a.eps) == (a0.lr, a0.betas, a0.eps) and a.t == a0.t
        g = a.grad if (grads is None or grads[i] is None) else grads[i]
        _need_device(g)
        assert g.numel() == a.param.numel()
        arr[i] = _abi.GhAdamTensor(a.param.data_ptr(), g.data_ptr(), a.exp_avg.data_ptr(), a.exp_avg_sq.data_ptr(), a.param.numel(),
                                   a.reg_l1, a.reg_l2, a.partials.data_ptr(), a.n_partials, a.step_state.data_ptr())
    with torch.cuda.device(a0.param.device):
        rc = L.gh_adam_reg_step_group(arr, len(adams), a0.t, a0.lr, a0.betas[0], a0.betas[1], a0.eps,
                                      None if guard is None else C.c_void_p(guard.data_ptr()), _stream(a0.param))
    if rc != 0:
        raise RuntimeError(f"gh_adam_reg_step_group failed: {_abi.status_name(rc)}")


def reg_total(a: AdamReg, col_a: int, k_a: float, b: AdamReg, col_b: int, k_b: float, base: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(base + reg, reg) with reg = k_a * sums(a)[col_a] + k_b * sums(b)[col_b] of the last steps' block partials, as one
    2-element device tensor from one small kernel (gh_reg_total)."""
    L = _lib.lib()
    out = torch.empty(2, dtype=torch.float32, device=a.param.device)
    if base is not None:
        base = base.detach().reshape(1)
        _need_device(base)
    with torch.cuda.device(out.device):
        rc = L.gh_reg_total(C.c_void_p(a.partials.data_ptr()), a.n_partials, int(col_a), float(k_a), C.c_void_p(b.partials.data_ptr()),
                            b.n_partials, int(col_b), float(k_b), None if base is None else C.c_void_p(base.data_ptr()),
                            C.c_void_p(out.data_ptr()), _stream(out))
    if rc != 0:
        raise RuntimeError(f"gh_reg_total failed: {_abi.status_name(rc)}")
    return out
