"""A short, fixed-seed run of the feature-matrix fuzz (tools/fuzz_features.py): random scene shapes x colour modes x blend subsets x
call variants (plain, split streams, static lists + refresh, shared geometry, occlusion bound, pose batch) against the C oracle —
forward bit-exact, gradients within the north star's tolerances (a float64 dense evaluation referees where two float32 programs
disagree on ill-conditioned scenes). The long runs are in profiles/r4_fuzz_features.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_feature_matrix_fuzz_fixed_seed():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_features.py"), "110", "7"], capture_output=True, text=True,
                       timeout=550, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-12:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("feature fuzz:")]
    assert summary, tail
    print(summary[0])
    assert "; 0 findings" in summary[0], tail
    # every variant was drawn
    import ast
    counts = ast.literal_eval(summary[0].split("): ", 1)[1].split("; ")[0])
    assert all(v > 0 for v in counts.values()), counts


@pytest.mark.timeout(600)
def test_dropin_sequence_fuzz_fixed_seed():
    """tools/fuzz_dropin.py: random sequences of drop-in calls (RGB + mask pass reuse, backwards in any order and long after later
    renders, in-place updates, replaced leaves, pool clears, all sync modes, learned instance capacities cut at random: overflow + recovery) — every image bit-equal to the oracle on the values the
    call saw, every gradient within tolerance, and a backward over inputs written in place since its forward raises autograd's error."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_dropin.py"), "60", "5", "--shrink"], capture_output=True, text=True, timeout=550, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-12:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("drop-in sequence fuzz:")]
    assert summary, tail
    print(summary[0])
    assert summary[0].endswith("; 0 findings"), tail
    import ast
    st = ast.literal_eval(summary[0].split("): ", 1)[1].rsplit("; ", 1)[0])
    assert st["mask_passes"] > 50 and st["late_backwards"] > 50 and st.get("stale_backwards", 0) > 10, st


@pytest.mark.timeout(600)
def test_fit_sequence_fuzz_fixed_seed():
    """tools/fuzz_fit.py: the one-shot fit in its three execution modes (static lists eager, static lists as replayed HIP graph, full
    path) through random schedules of steps, learning-rate milestones, moved Gaussians, dropped caches, pool clears and unrelated
    renders — eager and captured hold the same bits after every event, the full path agrees to float32 accumulation."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_fit.py"), "80", "6"], capture_output=True, text=True, timeout=550, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-12:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("fit sequence fuzz:")]
    assert summary, tail
    print(summary[0])
    assert "; 0 findings" in summary[0], tail


@pytest.mark.timeout(600)
def test_geometry_cache_sequence_fuzz_fixed_seed():
    """tools/fuzz_cache.py: `rasterize_views` through one shared GeometryCache while everything around it changes (in-place updates of
    what the static lists do and do not depend on, colour mode, SH degree, sizes, camera sets, blend terms, opacities lifted above
    the lists' culling bound, cleared caches and pools): every render bit-equal to the oracle, hit or build."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cache.py"), "25", "5"], capture_output=True, text=True, timeout=550, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-12:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("geometry-cache sequence fuzz:")]
    assert summary, tail
    print(summary[0])
    assert summary[0].endswith("; 0 findings"), tail
    import ast
    st = ast.literal_eval(summary[0].split("): ", 1)[1].rsplit("; ", 1)[0])
    assert st["hits"] > 60 and st["builds"] > 30, st


@pytest.mark.timeout(600)
def test_depth_bound_cache_sequence_fuzz_fixed_seed():
    """The same schedule through a DepthBoundCache on dense opaque scenes: whatever changed since the bound was reported (positions by
    centimetres, cameras, sizes, colour mode), a bounded call is the unbounded call bit for bit or a verified miss that is re-run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cache.py"), "16", "8", "--depth-bound"], capture_output=True, text=True,
                       timeout=550, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-12:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("depth-bound-cache sequence fuzz:")]
    assert summary, tail
    print(summary[0])
    assert summary[0].endswith("; 0 findings"), tail


@pytest.mark.timeout(600)
def test_size_class_fuzz_fixed_seed():
    """tools/fuzz_sizes.py: scenes drawn to select the code paths the launch logic picks BY SIZE (4 / 8 / 16 keys per thread in either
    sort, scan-free or row-scan histograms, three or four depth passes, 2 ... 17 tile bits, images up to 255 tiles wide, up to 5e7
    instances): image and radii bit-equal to the oracle's in every draw."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sizes.py"), "45", "2"], capture_output=True, text=True, timeout=550, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-8:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if l.startswith("size-class fuzz:")]
    assert summary, tail
    print(summary[0])
    assert summary[0].endswith("; 0 findings"), tail
    assert "tile-partition keys/thread [4, 8, 16]" in summary[0] and "depth-sort keys/thread [4, 8]" in summary[0], summary[0]
    assert "scan-free histogram [False, True]" in summary[0] and "depth24 [False, True]" in summary[0], summary[0]
