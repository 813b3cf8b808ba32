#!/usr/bin/env python3
"""Phase times of k_feature (per-feature stamps, 100 MHz ticks): K1, K2, K4, gate pass 1, pass 2, elimination."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth, _ffi
from msckf_amd.api import UpdateEngine
N, F, M = [int(x) for x in sys.argv[1:4]]
prob = synth.make_problem(N, F, M, seed=0)
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
    lib = e._lib
    lib.msckf_debug_fold_stamps(e._h, None, 0)
    e.load(prob)
    e.run(); e.run(); e.sync()
    n = min(F, 4096)
    out = (C.c_longlong * (8 * n))()
    got = lib.msckf_debug_fold_stamps(e._h, out, -n)
    a = np.frombuffer(out, dtype=np.int64).reshape(n, 8)[:got]
    d = np.diff(a, axis=1) * 10.0 / 1000.0     # us
    names = ["K1", "K2", "K4 write", "gate pass1", "gate pass2", "elimination", "tail"]
    print(f"N={N} F={F} M={M}: per-feature wave time {np.median(a[:, 7] - a[:, 0]) * 0.01:.1f} us (median)")
    for i, nm in enumerate(names):
        print(f"  {nm:12s} {np.median(d[:, i]):6.2f} us")
    span = (a[:, 7].max() - a[:, 0].min()) * 0.01
    print(f"  kernel span of the sampled features {span:.1f} us")
