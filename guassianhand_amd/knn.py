"""K nearest neighbours and the interaction mask — host side of gh_knn_* (include/gh_raster.h).

Reference call site, infer_one_shot.py:247-250 (knn_points = pytorch3d.ops.knn_points, a third-party dependency that
is not part of the reference tree):

    _, mink_idxs_world, _   = knn_points(pointclouds, pointclouds, K=100)
    _, mink_idxs_texture, _ = knn_points(t_point, t_point, K=100)
    mink_idxs_inter = (mink_idxs_world == mink_idxs_texture).sum(-1) < 10
    mink_idxs_inter = mink_idxs_inter.unsqueeze(-1)

`knn_points(p, p, K)` mirrors the reference call for the self-query case and returns the same 3-tuple shape
(dists (B,N,K), idx (B,N,K) int64, None); `interaction_mask` is the whole four-line block. ROCm tensors only: the
kernels raise if the library is missing and there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _abi, _lib


def _check(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("gh_knn_* runs on a ROCm device only (there is no CPU path)")
    if points.dim() != 3 or points.shape[-1] != 3:
        raise ValueError("points must be (B, N, 3)")
    return points.detach().float().contiguous()


def knn_indices(points: torch.Tensor, K: int, return_dists: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """points (B,N,3) -> idx (B,N,K) int32 sorted by (squared distance, index) [, dists (B,N,K)]."""
    p = _check(points)
    B, N, _ = p.shape
    L = _lib.lib()
    idx = torch.empty(B, N, K, dtype=torch.int32, device=p.device)
    dists = torch.empty(B, N, K, dtype=torch.float32, device=p.device) if return_dists else None
    nbytes = int(L.gh_knn_workspace_bytes(N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    with torch.cuda.device(p.device):
        stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
        for b in range(B):
            rc = L.gh_knn_indices(C.c_void_p(p[b].data_ptr()), N, K, C.c_void_p(idx[b].data_ptr()),
                                  C.c_void_p(dists[b].data_ptr()) if return_dists else None, C.c_void_p(ws.data_ptr()), nbytes, stream)
            if rc != 0:
                raise RuntimeError(f"gh_knn_indices failed: {_abi.status_name(rc)} (N={N}, K={K})")
    return idx, dists


def knn_points(p1: torch.Tensor, p2: torch.Tensor, K: int = 1):
    """Drop-in for the reference's two calls (self-query): returns (dists, idx int64, None) like pytorch3d's _KNN."""
    if p1 is not p2 and (p1.shape != p2.shape or p1.data_ptr() != p2.data_ptr()):
        raise NotImplementedError("only the self-query form knn_points(p, p, K) of infer_one_shot.py:247-248 is provided")
    idx, d = knn_indices(p1, K, return_dists=True)
    return d, idx.long(), None


def interaction_mask(pointclouds: torch.Tensor, t_point: torch.Tensor, K: int = 100, min_same: int = 10) -> torch.Tensor:
    """(B,N,3) posed points, (B,N,3) template-pose points -> (B,N,1) bool, True where fewer than `min_same` of the K
    sorted neighbour ranks agree (infer_one_shot.py:247-250)."""
    a, _ = knn_indices(pointclouds, K)
    b, _ = knn_indices(t_point, K)
    B, N, _ = a.shape
    L = _lib.lib()
    mask = torch.empty(B, N, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        for i in range(B):
            rc = L.gh_knn_mismatch_mask(C.c_void_p(a[i].data_ptr()), C.c_void_p(b[i].data_ptr()), N, K, min_same,
                                        C.c_void_p(mask[i].data_ptr()), stream)
            if rc != 0:
                raise RuntimeError(f"gh_knn_mismatch_mask failed: {_abi.status_name(rc)}")
    return mask.bool().unsqueeze(-1)
