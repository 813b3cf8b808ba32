#!/usr/bin/env python3
"""Per-wavefront cycle split of the fold step loop (first node of the LAST fold launch = the root; use
leaf-only problems to look at leaves)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (10, 8, 10)))
prob = synth.make_problem(N, F, M, seed=0)
eng = UpdateEngine(max_clones=N, max_features=F, max_track=M)
eng.load(prob)
eng._lib.msckf_debug_fold_stamps(eng._h, None, 0)
for _ in range(3):
    eng.run()
eng.sync()
buf = (C.c_longlong * 64)()
eng._lib.msckf_debug_fold_stamps(eng._h, buf, -1000000)
a = np.frombuffer(buf, dtype=np.int64)[:32].reshape(8, 4)
res = eng.result()
steps = 6 * N
print("levels", res.stats["n_levels"], "leaves", res.stats["n_leaves"])
print("wave: cycles/step  pre-barrier | barrier wait | sigma+rsq | k-loop")
for wv in range(8):
    print(wv, (a[wv] / steps).round(0).tolist(), "sum", round(float(a[wv].sum() / steps)))
