"""Gaussian selection of forward_single_batch (renderer_one_shot.py:468-473) at N = 98,562, C = 131: the reference's four
boolean-mask indexings against renderer.select_gaussians (gh_select_rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd.renderer import select_gaussians
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, C = 98562, 131
score, pts, feat = torch.rand(N, 1, generator=g).to(dev), torch.randn(N, 3, generator=g).to(dev), torch.randn(N, C, generator=g).to(dev)
def ref():
    s = score.squeeze(1)
    return pts[s > 0.1], feat[s > 0.1], pts[s > 0.9], feat[s > 0.9]
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a, b = t(ref), t(lambda: select_gaussians(score, pts, feat, 0.1, 0.9))
print(f"reference indexing (4 boolean masks): {a:.3f} ms;  select_gaussians: {b:.3f} ms   (N = {N}, C = {C})")
