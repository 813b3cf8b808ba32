#!/usr/bin/env python3
"""Host-side phases of the drop-in call (MSCKF_HOSTPROF=1 prints them when the engine closes): rotating batches at (N, F, M)."""
import os, sys, time
os.environ["MSCKF_HOSTPROF"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = (int(x) for x in (sys.argv[1:4] + ["30", "2000", "10"][len(sys.argv) - 1:]))
probs = [synth.make_problem(N, F, M, seed=sd) for sd in range(4)]
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
    for i in range(40): eng.update_problem(probs[i % 4])
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(200): eng.update_problem(probs[i % 4])
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    ms, _ = eng.run_timed(50)
    print("call %.1f us (best of 5 x 200), resident %.1f us" % (best, ms * 1000 / 50), flush=True)
