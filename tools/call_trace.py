#!/usr/bin/env python3
"""Wall time of every one of the first calls of the drop-in call in a fresh process (the driver's bench uses --steps 20 --warmup 5:
the timed region starts at call 5): usage: call_trace.py [calls] [N F M]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, F, M = (int(x) for x in (sys.argv[2:5] + ["30", "2000", "10"][len(sys.argv) - 2:])) if len(sys.argv) > 2 else (30, 2000, 10)
probs = [synth.make_problem(N, F, M, seed=sd) for sd in range(4)]
ts = []
with UpdateEngine(max_clones=N, max_features=F, max_track=max(M, 2)) as eng:
    for i in range(n):
        t0 = time.perf_counter()
        eng.update_problem(probs[i % 4])
        ts.append((time.perf_counter() - t0) * 1e6)
print(" ".join(f"{t:.0f}" for t in ts))
print("mean of calls 5..24: %.1f us; median of all: %.1f us" % (np.mean(ts[5:25]), np.median(ts)))
