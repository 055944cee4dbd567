"""cProfile of bench.two_call_cost's step (the reference protocol through the drop-in, one view per step, every leaf): own-time and
cumulative-time tables, per step."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from guassianhand_amd.scenes import make_scene

sc = make_scene("two_hands", n_views=8).to(torch.device("cuda:0"))
N = 200
pr = cProfile.Profile()
orig = bench.time.perf_counter
state = {"n": 0}


def hook():            # profile exactly the timed loop: enable at its first clock read, disable at its second
    state["n"] += 1
    if state["n"] == 1:
        pr.enable()
    elif state["n"] == 2:
        pr.disable()
    return orig()


bench.time.perf_counter = hook
ms = bench.two_call_cost(sc, [0], n_iter=N)
bench.time.perf_counter = orig
print(f"{ms:.3f} ms per view under cProfile")
for key in ("tottime", "cumtime"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(32)
    print(st.getvalue()[:7000])
