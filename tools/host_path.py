#!/usr/bin/env python3
"""Where the host-inclusive call (msckf_update: host arrays in, host arrays out) spends its time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (30, 2000, 10)))
prob = synth.make_problem(N, F, M, seed=0)
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
    for _ in range(5):
        res = eng.update_problem(prob)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); res = eng.update_problem(prob); ts.append(time.perf_counter() - t0)
    st = res.stats
    print(f"call {np.median(ts) * 1e6:.0f} us = device {st['us_total']:.0f} + host plan {st['us_host_prep']:.0f} + h2d {st['us_h2d']:.0f} "
          f"+ d2h {st['us_d2h']:.0f} + rest {np.median(ts) * 1e6 - st['us_total'] - st['us_host_prep'] - st['us_h2d'] - st['us_d2h']:.0f}")
