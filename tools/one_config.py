#!/usr/bin/env python3
"""Run one (N, F, M) config a few times on the resident path (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = [int(x) for x in sys.argv[1:4]]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dtype = sys.argv[5] if len(sys.argv) > 5 else "f64"
kind = sys.argv[6] if len(sys.argv) > 6 else "uniform"      # few: F - 10 ten-view + 10 M-view tracks; ragged: tracks ~ U[2, M]
if kind == "few":
    prob = synth.few_long_tracks_problem(N, F, 10, 10, seed=0)
elif kind == "ragged":
    prob = synth.make_problem(N, F, M, seed=0, variable_tracks=True, min_track=2)
else:
    prob = synth.make_problem(N, F, M, seed=0)
with UpdateEngine(max_clones=N, max_features=F, max_track=M, dtype=dtype) as eng:
    if "SPLIT_DIRECT_ROWS" in os.environ:          # remainder rows taken as they are up to this many (default 2048 / 16384)
        eng.set_rem_direct_rows(int(os.environ["SPLIT_DIRECT_ROWS"]))
    eng.load(prob)
    for _ in range(2):
        eng.run()
    ms, st = eng.run_timed(iters, stages=True)
    print(f"N={N} F={F} M={M} {dtype}: {ms / iters * 1000:.0f} us/update  stages {st}")
