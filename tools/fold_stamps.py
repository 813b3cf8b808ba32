#!/usr/bin/env python3
"""Diagnostic: per-node phase times of the fold kernel (wall_clock64 = 100 MHz ticks)."""
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
buf = (C.c_longlong * (8 * 4096))()
n = eng._lib.msckf_debug_fold_stamps(eng._h, buf, 4096)
a = np.frombuffer(buf, dtype=np.int64)[:8 * n].reshape(n, 8)
print("nodes", n)
# 100 MHz ticks -> us
us = a[:, :4] / 100.0
import collections
by_w = collections.defaultdict(list)
for i in range(n):
    by_w[(int(a[i, 4]), int(a[i, 5]) // 50 * 50)].append(us[i])
for k in sorted(by_w):
    v = np.array(by_w[k])
    print(f"w={k[0]:4d} rows~{k[1]:4d} n={len(v):4d}  setup={v[:,0].mean():7.1f}  staging={v[:,1].mean():7.1f}  steps={v[:,2].mean():7.1f}  total={v[:,3].mean():7.1f} us")
fine = a[:, 6:8].astype(np.uint64)
for i in list(range(0, 3)) + list(range(n - 4, n)):
    cA, cB = int(fine[i, 0]) >> 32, int(fine[i, 0]) & 0xffffffff
    cC, cD = int(fine[i, 1]) >> 32, int(fine[i, 1]) & 0xffffffff
    steps = max(1, int(a[i, 4]))
    print(f"node {i}: w={int(a[i,4])} rows={int(a[i,5])} cycles/step: to-barrier={cA/steps:.0f} vread+sigma={cB/steps:.0f} rsq={cC/steps:.0f} kloop={cD/steps:.0f}  total={(cA+cB+cC+cD)/steps:.0f}")
