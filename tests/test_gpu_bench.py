"""bench.py's multi-GPU leg, exercised before a driver ever runs it on an 8-GPU node: the sharded code path at world
size 1 (`--force-dist`: id file, stdout kept for the ONE JSON line, strong + weak cases, the one-GPU reference run) in
a fresh interpreter, and the default one-GPU line's shape."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, timeout=600):
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", "29531")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print ONE line on stdout, got %d: %r" % (len(lines), lines[:3])
    return json.loads(lines[0])


def test_sharded_leg_at_world_one():
    j = _run_bench(["--force-dist", "--steps", "3", "--warmup", "1"])
    assert j["status"] == [0, 0]
    assert j["scaling"] == "strong" and j["n_gpus"] == 1
    assert j["accepted"] + j["rejected"] == 8000
    assert j["value"] > 0 and j["ms_per_step"] > 0
    assert j["weak_scaling"]["scaling"] == "weak" and j["weak_scaling"]["updates_per_s"] > 0
    assert j["one_gpu_same_workload"]["us_per_update"] > 0
    assert j["config"]["exchange"] == "group triangles"


def test_one_gpu_line_shape():
    j = _run_bench(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extra-configs"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["dtype"] == "f64" and j["scaling"] == "none"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert 0.0 < r["frac"] < 1.0 and r["achieved"] > 0 and r["peak"] > 0
