#!/usr/bin/env python3
"""Race / hang screen: a few dozen different batches (window sizes, ragged and consecutive tracks, with and without wide-class
tracks, 200 - 6000 tracks) in rotation through the one-shot call for `calls` calls; every result must equal, bit for bit, the
first result of its batch, and no call may report a timeout of the in-launch waits (MSCKF_ERR_HIP).
usage: stress_repeat.py [calls] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from soak_holes import ragged

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
batches = []
for k in range(24):
    N = int(rng.choice([8, 16, 20, 30, 30, 30, 40, 50]))
    F = int(rng.choice([200, 500, 2000, 2000, 4000, 6000]))
    if os.environ.get("STRESS_ONLY") == "short":         # short ragged tracks whose holes make them span > 10 slots (wide class)
        batches.append(ragged(rng, 30, 1500, 2, int(rng.integers(3, 7)), 0.1))
    elif os.environ.get("STRESS_ONLY") == "ragged":
        batches.append(ragged(rng, 30, 1500, 2, int(rng.integers(2, 31)), 0.1))
    elif os.environ.get("STRESS_ONLY") == "plain":
        batches.append(synth.make_problem(30, 2000, 10, seed=int(rng.integers(1 << 30)), outlier_fraction=0.1, outlier_px=300.0))
    elif k % 3 == 0:
        batches.append(ragged(rng, N, min(F, 1500), 2, int(rng.integers(2, min(N, 31) + 1)), 0.1))
    else:
        batches.append(synth.make_problem(N, F, int(rng.integers(3, min(N, 15) + 1)), seed=int(rng.integers(1 << 30)),
                                          variable_tracks=bool(rng.integers(2)), outlier_fraction=0.1, outlier_px=300.0))
first = [None] * len(batches)
t0 = time.time()
with UpdateEngine(max_clones=53, max_features=6000, max_track=31, plan=os.environ.get("STRESS_PLAN", "auto")) as eng:
    for i in range(calls):
        b = int(rng.integers(len(batches)))
        r = eng.update_problem(batches[b])
        if first[b] is None:
            first[b] = r
        elif not (np.array_equal(r.dx, first[b].dx) and np.array_equal(r.P_new, first[b].P_new) and np.array_equal(r.accepted, first[b].accepted)):
            Ms = np.diff(batches[b].view_ptr)
            print(f"call {i}: batch {b} (N={batches[b].N}, F={batches[b].F}, longest track {Ms.max()}) differs from its first result: dx {np.abs(r.dx - first[b].dx).max():.3e} "
                  f"P {np.abs(r.P_new - first[b].P_new).max():.3e} masks equal {np.array_equal(r.accepted, first[b].accepted)} status {r.status}/{first[b].status}", flush=True)
            nbad = globals().get("nbad", 0) + 1; globals()["nbad"] = nbad
            if nbad >= 3: sys.exit(1)
        if i % 5000 == 4999:
            print(f"{i + 1} calls, {time.time() - t0:.0f} s", flush=True)
print(f"{calls} calls over {len(batches)} batches: all results bitwise equal to the first of their batch, no error")
