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

# feature-kernel phases (100 MHz ticks)
nf = eng._lib.msckf_debug_fold_stamps(eng._h, buf, -1024)
fa = np.frombuffer(buf, dtype=np.int64)[:8 * nf].reshape(nf, 8)
d = np.diff(fa, axis=1) / 100.0
names = ["K1 rows", "K2 QR+Z", "K4 store", "gate pass1 (E, ZP)", "gate pass2 (S)", "elimination", "epilogue"]
print("k_feature phases, mean us over", nf, "features:", ", ".join(f"{n}={d[:, i].mean():.1f}" for i, n in enumerate(names)),
      f" total={(fa[:, 7] - fa[:, 0]).mean() / 100.0:.1f}")
nf = eng._lib.msckf_debug_fold_stamps(eng._h, buf, -8192)
fa = np.frombuffer(buf, dtype=np.int64)[:8 * nf].reshape(nf, 8)
ev = sorted([(int(r[0]), 1) for r in fa] + [(int(r[7]), -1) for r in fa])
cur = mx = 0
for _, dlt in ev:
    cur += dlt; mx = max(mx, cur)
t0 = fa[:, 0].min()
print("features", nf, "max concurrent blocks", mx, "first start..last start", (fa[:, 0].max() - t0) / 100.0, "us; last end", (fa[:, 7].max() - t0) / 100.0, "us")
hist = np.histogram((fa[:, 0] - t0) / 100.0, bins=8)
print("start-time histogram (us):", [f"{b:.0f}" for b in hist[1]], hist[0].tolist())
d = np.diff(fa, axis=1) / 100.0
tot = (fa[:, 7] - fa[:, 0]) / 100.0
print("per-phase p50 / p95 / max (us):")
for i, nme in enumerate(names):
    print(f"  {nme:22s} {np.percentile(d[:, i], 50):6.1f} {np.percentile(d[:, i], 95):6.1f} {d[:, i].max():6.1f}")
print(f"  {'total':22s} {np.percentile(tot, 50):6.1f} {np.percentile(tot, 95):6.1f} {tot.max():6.1f}")
order = np.argsort(tot)[-5:]
print("slowest features:", order.tolist(), "their phases:", np.round(d[order], 1).tolist())
print("mean total by feature-index decile:", [round(float(tot[i::10].mean()), 1) for i in range(1)], [round(float(tot[k * 200:(k + 1) * 200].mean()), 1) for k in range(10)])
