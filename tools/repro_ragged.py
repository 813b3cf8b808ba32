#!/usr/bin/env python3
"""Re-runs one batch of tests/test_gpu_parity.py::test_ragged_tracks_soak (index argv[1]) with diagnostics."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
spec = importlib.util.spec_from_file_location("soak_holes", os.path.join(ROOT, "tools", "soak_holes.py"))
sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
want = int(sys.argv[1])
rng = np.random.default_rng(11)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
for c in range(want + 1):
    N = int(rng.integers(2, 54)); F = int(rng.integers(1, 200))
    hi = int(rng.integers(2, min(N, 31) + 1))
    prob = sh.ragged(rng, N, F, 2, hi, float(rng.choice([0.0, 0.1, 0.4])))
print("case", want, "N", N, "F", F, "hi", hi)
ref = oracle.update(prob, dense_noise=False)
H, r = ref["H_X"][:, 15:], ref["r_o"]
for direct in (-1, 0):
    with UpdateEngine(max_clones=53, max_features=2048, max_track=31) as eng:
        eng.set_rem_direct_rows(direct)
        for rep in range(3):
            res = eng.update_problem(prob)
            T, rn = eng.debug_compressed()
            print("direct", direct, "rep", rep, eng.debug_split(), "dx", rel(res.dx, ref["dx"]), "P", rel(res.P_new, ref["P_new"]),
                  "TtT", rel(T.T @ T, H.T @ H), "Ttr", rel(T.T @ rn, H.T @ r), "mask", np.array_equal(res.accepted, ref["accepted"]))
