#!/usr/bin/env python3
"""Randomised parity soak: many (N, F, M) shapes against the oracle through the one-shot call, both dtypes' tolerances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
worst = (0.0, 0.0)
for ci in range(ncase):
    N = int(rng.integers(2, 54))
    M = int(rng.integers(2, min(N, 16) + 1))
    F = int(rng.integers(1, 400))
    kw = {}
    if rng.random() < 0.4: kw["variable_tracks"] = True
    if rng.random() < 0.3: kw.update(outlier_fraction=0.1, outlier_px=300.0)
    seed = int(rng.integers(0, 10**6))
    prob = synth.make_problem(N, F, M, seed=seed, **kw)
    ref = oracle.update(prob, dense_noise=False)
    with UpdateEngine(max_clones=N, max_features=F, max_track=max(M, 2)) as e:
        r = e.update_problem(prob)
    ok = r.status == ref["status"] and np.array_equal(r.accepted, ref["accepted"])
    edx = eP = 0.0
    if ok and ref["status"] == 0:
        edx = np.linalg.norm(r.dx - ref["dx"]) / max(np.linalg.norm(ref["dx"]), 1e-300)
        eP = np.linalg.norm(r.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"])
        ok = edx < 1e-8 and eP < 1e-8
        worst = (max(worst[0], edx), max(worst[1], eP))
    if not ok:
        bad += 1
        print("FAIL", dict(N=N, F=F, M=M, seed=seed, **kw), r.status, ref["status"], edx, eP, flush=True)
print(f"soak: {ncase} cases, {bad} failures, worst dx {worst[0]:.2e} P {worst[1]:.2e}")
sys.exit(1 if bad else 0)
