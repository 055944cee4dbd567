"""Time the device kNN (K=100) and the interaction mask on the full two-hand cloud."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import knn
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=1)
p = sc.xyz[None].to(dev)
tp = (sc.xyz * torch.tensor([1.0, 1.0, 1.0]) + torch.where(torch.arange(sc.P)[:, None] < sc.P // 2, 0.0, 0.3))[None].to(dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"knn_indices K=100, N={sc.P}: {t(lambda: knn.knn_indices(p, 100)):.3f} ms")
print(f"interaction_mask (2x kNN + compare): {t(lambda: knn.interaction_mask(p, tp)):.3f} ms")
m = knn.interaction_mask(p, tp)
print("flagged fraction:", float(m.float().mean()))
