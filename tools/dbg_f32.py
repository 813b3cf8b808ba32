#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
for (N, F, M) in [(20, 500, 8), (30, 2000, 10), (50, 2000, 15)]:
    prob = synth.make_problem(N, F, M, seed=0)
    ref = oracle.update(prob, dense_noise=False)
    for dt in ("f64", "f32"):
        with UpdateEngine(max_clones=N, max_features=F, max_track=M, dtype=dt) as e:
            r = e.update_problem(prob)
            print(N, F, M, dt, "dx", np.linalg.norm(r.dx - ref["dx"]) / np.linalg.norm(ref["dx"]), "P", np.linalg.norm(r.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"]),
                  "acc eq", np.array_equal(r.accepted, ref["accepted"]))
