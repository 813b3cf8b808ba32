#!/usr/bin/env python3
"""Diagnostic: a list of (N, F, M) problems through the drop-in call against the oracle; prints the K5 plan and the errors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
cases = [(10, 1, 5, 3, {}), (10, 2, 5, 3, {}), (10, 6, 5, 3, {}), (10, 12, 5, 3, {}), (10, 24, 5, 3, {}), (10, 6, 10, 3, {}), (3, 2, 3, 3, {}), (10, 50, 5, 3, dict(outlier_fraction=0.1, outlier_px=500.0)), (10, 50, 5, 3, {}), (10, 50, 10, 3, {}), (10, 200, 5, 3, {}),
         (12, 60, 6, 1, {}), (30, 500, 10, 1, {}), (20, 100, 14, 1, {}), (40, 300, 10, 1, {}), (10, 50, 4, 2, {}), (10, 50, 8, 2, {})]
for (N, F, M, sd, kw) in cases:
    prob = synth.make_problem(N, F, M, seed=sd, **kw)
    ref = oracle.update(prob, dense_noise=False)
    with UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
        r = e.update_problem(prob)
        rule = e.band_rule(N, M) if hasattr(e, "band_rule") else -1
    edx = np.linalg.norm(r.dx - ref["dx"]) / np.linalg.norm(ref["dx"])
    eP = np.linalg.norm(r.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"])
    print(N, F, M, kw, "rule", rule, "status", r.status, ref["status"], "acc", int(r.accepted.sum()), int(ref["accepted"].sum()), "dx %.2e P %.2e" % (edx, eP), flush=True)
