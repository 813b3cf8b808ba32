#!/usr/bin/env python3
"""Repeats ragged long-track batches on ONE engine and compares every result with the oracle (and bit for bit with the batch's
first result).  usage: stress_split.py [rounds] [batches]   env: the MSCKF_* switches, SPLIT_DIRECT_ROWS"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
spec = importlib.util.spec_from_file_location("soak_holes", os.path.join(ROOT, "tools", "soak_holes.py"))
sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(11)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
probs, refs = [], []
while len(probs) < nb:
    N = int(rng.integers(11, 54)); F = int(rng.integers(20, 200))
    hi = int(rng.integers(11, min(N, 31) + 1))
    p = sh.ragged(rng, N, F, 2, hi, float(rng.choice([0.0, 0.1, 0.4])))
    r = oracle.update(p, dense_noise=False)
    if r["status"] == 0:
        probs.append(p); refs.append(r)
first = [None] * nb
bad = 0
t0 = time.time()
with UpdateEngine(max_clones=53, max_features=2048, max_track=31) as eng:
    if "SPLIT_DIRECT_ROWS" in os.environ:
        eng.set_rem_direct_rows(int(os.environ["SPLIT_DIRECT_ROWS"]))
    calls = 0
    for rd in range(rounds):
        if rd and rd % 500 == 0:
            print(f"... {calls} calls, {bad} bad, {time.time() - t0:.0f} s", flush=True)
        order = np.random.default_rng(rd).permutation(nb)
        for i in order:
            res = eng.update_problem(probs[i])
            calls += 1
            e = max(rel(res.dx, refs[i]["dx"]), rel(res.P_new, refs[i]["P_new"]))
            same = first[i] is None or (np.array_equal(res.dx, first[i][0]) and np.array_equal(res.P_new, first[i][1]))
            if first[i] is None and e < 1e-8:
                first[i] = (res.dx.copy(), res.P_new.copy())
            if e > 1e-8 or not same or res.status != 0:
                bad += 1
                s = eng.debug_split()
                print(f"round {rd} batch {i} N {probs[i].N} F {probs[i].F} status {res.status} err {e:.2e} bitwise {same} split {s}", flush=True)
print(f"{calls} calls, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
