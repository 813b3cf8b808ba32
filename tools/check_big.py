#!/usr/bin/env python3
"""One-off parity + timing at the large BASELINE.json configs against the oracle (slow on the CPU side)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle

for (N, F, M) in [(30, 8000, 10), (30, 10000, 10), (50, 20000, 15)]:
    t0 = time.time()
    prob = synth.make_problem(N, F, M, seed=0)
    t_gen = time.time() - t0
    with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
        res = eng.update_problem(prob)
        eng.load(prob)
        for _ in range(2):
            eng.run()
        ms, st = eng.run_timed(10, stages=True)
    print(f"N={N} F={F} M={M}: status={res.status} accepted={int(res.accepted.sum())} device {ms/10*1000:.0f} us/update "
          f"(feature {st[0]:.0f}, qr {st[1]:.0f}, gain {st[2]:.0f}) -> {10000/ms:.1f} updates/s; leaves={res.stats['n_leaves']} "
          f"levels={res.stats['n_levels']} gen {t_gen:.0f}s", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "--oracle":
        t0 = time.time()
        ref = oracle.update(prob, dense_noise=False)
        dt = time.time() - t0
        e_dx = np.linalg.norm(res.dx - ref["dx"]) / np.linalg.norm(ref["dx"])
        e_P = np.linalg.norm(res.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"])
        print(f"    oracle {dt:.1f} s; accepted equal={np.array_equal(res.accepted, ref['accepted'])} dx_rel={e_dx:.2e} P_rel={e_P:.2e}", flush=True)
