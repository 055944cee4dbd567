"""bench.py's launch logic without a GPU: a rank count that does not match --gpus must fail loudly, and
`python bench.py --gpus N` without a launcher must start N ranks itself (here each of them stops at "needs a ROCm device",
which is the proof that they were started, with the right world size, before any GPU call)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_rank_count_mismatch_fails_loudly():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert "--gpus 8 but WORLD_SIZE is 1" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no JSON line that could be taken for an 8-GPU result


def test_self_launch_starts_the_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=_env(), timeout=600)
    assert r.returncode != 0                                                  # no GPU here: every rank refuses to run
    assert r.stderr.count("bench.py needs a ROCm device") == 2, r.stderr[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
