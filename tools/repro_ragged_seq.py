#!/usr/bin/env python3
"""The 60 batches of tests/test_gpu_parity.py::test_ragged_tracks_soak on ONE engine, with diagnostics per batch."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
spec = importlib.util.spec_from_file_location("soak_holes", os.path.join(ROOT, "tools", "soak_holes.py"))
sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
rng = np.random.default_rng(11)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
with UpdateEngine(max_clones=53, max_features=2048, max_track=31) as eng:
    for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        N = int(rng.integers(2, 54)); F = int(rng.integers(1, 200))
        hi = int(rng.integers(2, min(N, 31) + 1))
        prob = sh.ragged(rng, N, F, 2, hi, float(rng.choice([0.0, 0.1, 0.4])))
        ref = oracle.update(prob, dense_noise=False)
        res = eng.update_problem(prob)
        s = eng.debug_split()
        e1 = rel(res.dx, ref["dx"]) if ref["status"] == 0 else 0.0
        line = f"case {c} N {N} F {F} hi {hi} status {res.status}/{ref['status']} mask {np.array_equal(res.accepted, ref['accepted'])} dx {e1:.1e} long {s['long_tracks']} narrow {s['narrow_blocks']} cap {s['remainder_rows_cap']} mode {s['remainder_mode']} band {s['band_plan']} sweep {s['sweep_mode']}"
        if e1 > 1e-8:
            T, rn = eng.debug_compressed()
            H, r = ref["H_X"][:, 15:], ref["r_o"]
            line += f" TtT {rel(T.T @ T, H.T @ H):.1e}"
            res2 = eng.update_problem(prob)
            line += f" again dx {rel(res2.dx, ref['dx']):.1e}"
        print(line, flush=True)
