"""kNN oracle (oracle/gh_oracle.c:gho_knn) against an independent numpy restatement of pytorch3d's knn_points as the
reference calls it (infer_one_shot.py:247-248: self-query, K=100, sorted) and the mask arithmetic of :249."""
import numpy as np
import torch

from oracle import oracle_c


def numpy_knn(p: np.ndarray, K: int):
    """Dense restatement: squared L2 in float32 accumulated x, y, z with fused multiply-adds, lexsort by (distance, index)."""
    d = p[:, None, :].astype(np.float64) - p[None, :, :].astype(np.float64)          # differences are exact in f64
    d32 = d.astype(np.float32)                                                        # == the f32 subtraction (correctly rounded)
    acc = (d32[..., 0].astype(np.float64) ** 2).astype(np.float32)                    # dx*dx rounded to f32
    acc = (d32[..., 1].astype(np.float64) ** 2 + acc.astype(np.float64)).astype(np.float32)   # fma: one rounding (exact in f64)
    acc = (d32[..., 2].astype(np.float64) ** 2 + acc.astype(np.float64)).astype(np.float32)
    idx = np.lexsort((np.broadcast_to(np.arange(p.shape[0]), acc.shape), acc), axis=-1)[:, :K]
    return idx.astype(np.int32), np.take_along_axis(acc, idx, axis=-1)


def test_oracle_knn_matches_dense_numpy_restatement():
    g = torch.Generator().manual_seed(5)
    p = torch.rand(700, 3, generator=g)
    p[100:110] = p[0:10]                       # exact duplicates: distance ties, resolved by index
    p[200:300, 2] = 0.5                        # a coplanar patch
    idx, d = oracle_c.knn(p, 100)
    want_idx, want_d = numpy_knn(p.numpy(), 100)
    assert np.array_equal(idx.numpy(), want_idx)
    assert np.array_equal(d.numpy(), want_d)
    assert np.array_equal(idx[:, 0].numpy()[:100], np.arange(100))      # self first (for duplicates: the lower index)
    assert (idx[100:110, 0].numpy() == np.arange(10)).all()


def test_oracle_knn_query_subset_and_mask():
    g = torch.Generator().manual_seed(6)
    p = torch.randn(500, 3, generator=g)
    full, _ = oracle_c.knn(p, 32)
    qs = torch.tensor([3, 499, 17, 3], dtype=torch.int32)
    sub, _ = oracle_c.knn(p, 32, queries=qs)
    assert torch.equal(sub, full[qs.long()])
    # rigid motion keeps every neighbour list -> nothing flagged; a shuffled second set flags nearly everything
    R = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    moved = p.double() @ R.double().T + 0.25
    m_same = oracle_c.interaction_mask(p, p.clone(), K=32, min_same=10)
    assert not m_same.any()
    m_rigid = oracle_c.interaction_mask(p, moved.float(), K=32, min_same=10)
    assert m_rigid.float().mean() < 0.05                                # only rounding-induced rank swaps
    m_diff = oracle_c.interaction_mask(p, torch.randn(500, 3, generator=g), K=32, min_same=10)
    assert m_diff.float().mean() > 0.95
