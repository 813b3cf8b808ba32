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


def _run_bench(extra, timeout=600, max_len=None):
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", "29531")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print ONE line on stdout, got %d: %r" % (len(lines), lines[:3])
    if max_len is not None:
        assert len(lines[0]) < max_len, len(lines[0])
    return json.loads(lines[0])


def test_sharded_leg_at_world_one():
    j = _run_bench(["--force-dist", "--steps", "3", "--warmup", "1"], max_len=4096)
    assert j["status"] == [0, 0]
    assert j["scaling"] == "strong" and j["n_gpus"] == 1
    assert j["accepted"] + j["rejected"] == 8000
    assert j["value"] > 0 and j["ms_per_step"] > 0
    assert j["weak_scaling"]["scaling"] == "weak" and j["weak_scaling"]["updates_per_s"] > 0
    assert j["one_gpu_same_workload"]["us_per_update"] > 0
    assert j["config"]["exchange"] == "group triangles"


def test_the_drivers_command_prints_one_short_line():
    """The command the driver runs at round end, no extra flags: ONE stdout line under 4 KB (round 4's 20 KB line came back
    unparsed) with `roofline` and `cpu_baseline`; the detail lands in bench_detail.json."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), [len(l) for l in lines])
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "north_star_roofline"):
        assert key in j, key
    assert j["steps"] == 20 and j["warmup"] == 5 and j["n_gpus"] == 1
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in j["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in j["cpu_baseline"], key
    assert len(j["cpu_baseline"]["sample"]) <= 200
    assert j["parity_vs_cpu_baseline"]["dx_rel"] < 1e-8 and j["parity_vs_cpu_baseline"]["P_rel"] < 1e-8
    assert j["north_star_roofline"]["frac"] > 0.3
    d = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert len(d["configs"]) >= 8 and not [r for r in d["configs"] if "error" in r], [r for r in d["configs"] if "error" in r]


def test_one_gpu_line_shape():
    j = _run_bench(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extra-configs"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["dtype"] == "f64" and j["scaling"] == "none"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert 0.0 < r["frac"] < 1.0 and r["achieved"] > 0 and r["peak"] > 0
