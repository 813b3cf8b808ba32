#!/usr/bin/env python3
"""Diagnostic (library built with -DSWEEP_PROF): shader-clock cycles per phase of the root sweep's steps."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (30, 2000, 10)))
prob = synth.make_problem(N, F, M, seed=0)
eng = UpdateEngine(max_clones=N, max_features=F, max_track=M)
eng.load(prob)
eng._lib.msckf_debug_fold_stamps(eng._h, None, 0)
for _ in range(3):
    eng.run()
eng.sync()
buf = (C.c_longlong * 64)()
eng._lib.msckf_debug_fold_stamps(eng._h, buf, -1000000)
a = np.frombuffer(buf, dtype=np.int64).reshape(8, 8)
names = ["idle", "dots", "reduce", "scalars+tau+R", "update", "barrier", "tau hand-back"]
print("per-wave cycles per active step (root sweep); idle = waiting for the fold's first step / the other wavefronts' last ones")
for w in range(8):
    steps = max(1, int(a[w, 7]))
    print(f"wave {w}: steps={steps:4d} " + "  ".join(f"{n}={a[w, i] / steps:6.0f}" for i, n in enumerate(names) if n != "-"), f" total/step={a[w, 1:7].sum() / steps:.0f}  idle total={a[w, 0]:.0f}")
