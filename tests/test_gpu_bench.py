"""bench.py contract (driver-facing): one JSON line with the agreed fields, on one GPU and through the N > 1 control
path (two ranks on the one card with the gloo backend — the collectives are real, the numbers are not)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_one_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5"],
                       capture_output=True, text=True, cwd=ROOT, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["value"] > 300.0                                   # BASELINE.json's floor for this workload
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]      # renders / wall time
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.0 < rf["frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.timeout(600)
def test_bench_two_ranks_control_path():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dist-backend", "gloo", "--no-stage-timing"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["collective"] == "all-reduce(loss)" and d["cpu_baseline"] is None
    assert d["config"]["parallelism"] == "view-parallel x2"
    # the reduced loss is the sum of both ranks' losses (two different sets of 8 views): well above a single rank's
    assert d["config"]["final_loss"] > 0.03
