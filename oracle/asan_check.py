"""Runs the oracle's forward, backward and kNN on small scenes against the ASan/UBSan build (see oracle/Makefile: asan)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle_c
oracle_c._LIB_PATH = os.path.join(ROOT, "oracle", "libgh_oracle_asan.so")
oracle_c._lib = None
from guassianhand_amd.scenes import make_scene
from tests.helpers import dimg_like, scene_kwargs
for par, cfg, kw in ((0, "random1k", dict(n_views=2, blend=True)), (0, "random1k", dict(n_views=1, use_rgb=False, blend=True)),
                     (0, "one_hand", dict(n_views=1, P=5000)),
                     (1, "random1k", dict(n_views=2, blend=True)), (1, "one_hand", dict(n_views=2, P=5000, use_rgb=False))):   # baseline mode
    oracle_c.set_parallel(bool(par))
    sc = make_scene(cfg, **kw)
    k, bl = scene_kwargs(sc)
    o = oracle_c.OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=True, **k, **bl)
    g = o.backward(dimg_like(sc.w2c.shape[0], sc.H, sc.W))
    o.close()
    print("baseline mode" if par else "checker", cfg, kw, "instances", o.num_rendered, "sum|dL/dxyz|", float(g["means3D"].abs().sum()))
idx, d = oracle_c.knn(torch.rand(3000, 3), 100)
print("knn", tuple(idx.shape), "clean")
