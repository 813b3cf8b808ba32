#!/usr/bin/env python3
"""Long tracks through the split (two-level nullspace basis, k_feature.h) against the oracle: a quick screen on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

cases = [
    ("few long among short", lambda: synth.few_long_tracks_problem(30, 400, 10, 10, seed=42)),
    ("31 x 64 x 31", lambda: synth.make_problem(31, 64, 31, seed=24)),
    ("U[2,30] + outliers", lambda: synth.make_problem(30, 300, 30, seed=41, variable_tracks=True, min_track=2, outlier_fraction=0.1, outlier_px=400.0)),
    ("U[2,15]", lambda: synth.make_problem(30, 300, 15, seed=5, variable_tracks=True, min_track=2)),
    ("16 x 80 x 16", lambda: synth.make_problem(16, 80, 16, seed=55)),
    ("N=53 long", lambda: synth.make_problem(53, 200, 31, seed=7, variable_tracks=True, min_track=2)),
    ("N=12, 12 views", lambda: synth.make_problem(12, 80, 12, seed=4, variable_tracks=True, min_track=2)),
    ("only long", lambda: synth.make_problem(30, 40, 30, seed=9, variable_tracks=True, min_track=16)),
    ("one long", lambda: synth.make_problem(20, 1, 20, seed=3)),
]
worst = 0.0
with UpdateEngine(max_clones=53, max_features=4096, max_track=31) as eng:
    for name, mk in cases:
        prob = mk()
        ref = oracle.update(prob, dense_noise=False)
        res = eng.update_problem(prob)
        ok = res.status == ref["status"] and np.array_equal(res.accepted, ref["accepted"])
        e1, e2 = rel(res.dx, ref["dx"]), rel(res.P_new, ref["P_new"])
        gam, q = eng.debug_gate()
        eg = float(np.max(np.abs(gam - ref["gamma"]) / np.maximum(np.abs(ref["gamma"]), 1e-12)))
        T, rn = eng.debug_compressed()
        G = T.T @ T
        Gr = ref["H_X"][:, 15:].T @ ref["H_X"][:, 15:]
        print(f"{name:24s} status {res.status}/{ref['status']} mask {ok} dx {e1:.2e} P {e2:.2e} gamma {eg:.1e} TtT {rel(G, Gr):.1e} rows {res.stats['stacked_rows']} us {res.stats['us_total']:.0f}", flush=True)
        worst = max(worst, e1, e2)
        # resident repeat
        eng.load(prob)
        for _ in range(3):
            eng.run()
        eng.sync()
        r2 = eng.result()
        print(f"{'':24s} resident dx {rel(r2.dx, ref['dx']):.2e} P {rel(r2.P_new, ref['P_new']):.2e}")
        ms, st = eng.run_timed(20, stages=True)
        print(f"{'':24s} {1e3 * ms / 20:.0f} us/update stages {[round(x) for x in st]}")
print("worst", worst)
sys.exit(0 if worst < 1e-8 else 1)
