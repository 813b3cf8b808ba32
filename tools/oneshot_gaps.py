#!/usr/bin/env python3
"""Kernel gaps inside the one-shot call (run under rocprofv3 --kernel-trace): rotating batches at (N, F, M)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = (int(x) for x in (sys.argv[1:4] + ["30", "2000", "10"][len(sys.argv) - 1:]))
probs = [synth.make_problem(N, F, M, seed=sd) for sd in range(4)]
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
    for i in range(60):
        eng.update_problem(probs[i % 4])
