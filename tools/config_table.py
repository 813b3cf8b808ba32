#!/usr/bin/env python3
"""Timing table over BASELINE.json's configs and frame-sized batches (markdown rows on stdout).
Resident rate: inputs in HBM, K1..K7 (HIP events). Host-inclusive: msckf_update (host arrays in/out)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine

CONFIGS = [(10, 50, 5), (20, 500, 8), (30, 100, 10), (30, 300, 10), (30, 2000, 10), (30, 8000, 10), (30, 10000, 10),
           (50, 20000, 15)]
print("| N | F | M | device us/update | updates/s | K1-K4 | K5 | K6-K7 | levels | host-inclusive updates/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (N, F, M) in CONFIGS:
    prob = synth.make_problem(N, F, M, seed=0)
    with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
        res = eng.update_problem(prob)
        t0 = time.perf_counter()
        reps = 10 if F <= 2000 else 3
        for _ in range(reps):
            eng.update_problem(prob)
        host = reps / (time.perf_counter() - t0)
        eng.load(prob)
        for _ in range(3):
            eng.run()
        it = 50 if F <= 2000 else 10
        ms, st = eng.run_timed(it, stages=True)
    us = ms / it * 1000
    print(f"| {N} | {F} | {M} | {us:.0f} | {1e6 / us:.0f} | {st[0]:.0f} | {st[1]:.0f} | {st[2]:.0f} | "
          f"{res.stats['n_levels']} | {host:.0f} |", flush=True)
