"""The CPU oracle on this host's cores: its checker mode (emit / sort / chain rule on one thread) against its baseline mode (every
stage under OpenMP), library seconds per forward + backward render of view 0 of the default workload."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from guassianhand_amd.scenes import make_scene
from oracle import oracle_c
s = make_scene("two_hands", n_views=8)
cams = s.cams()[:1]
dimg = torch.randn(1, 3, s.H, s.W) / (3 * s.H * s.W)
kw = dict(colors_precomp=s.shs.squeeze(1))
print("cpu:", oracle_c.cpu_model(), "threads:", oracle_c.num_threads(), "host cpus:", os.cpu_count())
for mode in (False, True):
    oracle_c.set_parallel(mode)
    for rep in range(2):
        oracle_c.timing(reset=True)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 6.0:
            r = oracle_c.OracleRender(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, xyz_b=s.xyz_b, opacity_b=s.opacity_b,
                                      color_w=s.color_w, color_b=s.color_b, **kw)
            r.backward(dimg); r.close(); n += 1
        wall = time.perf_counter() - t0
        lib, ser = oracle_c.timing()
    print(f"{'baseline mode (all stages parallel)' if mode else 'checker mode (serial emit/sort/chain)'}: {n / wall:6.2f} renders/s by wall clock, "
          f"{n / lib:6.2f} by library time ({1e3 * lib / n:.0f} ms per render, {100 * ser / lib:.1f} % of it on one thread)")
oracle_c.set_parallel(False)
