import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
bad = 0
for (N, F, M, seeds) in [(16, 120, 14, range(81, 89)), (20, 400, 15, range(5)), (12, 300, 11, range(3)), (30, 2000, 10, range(2))]:
    with UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
        for sd in seeds:
            prob = synth.make_problem(N, F, M, seed=sd)
            ref = oracle.update(prob, dense_noise=False)
            first = None
            for it in range(40):
                r = e.update_problem(prob)
                if first is None:
                    first = r
                    edx = np.linalg.norm(r.dx - ref["dx"]) / np.linalg.norm(ref["dx"])
                    if edx > 1e-8 or not np.array_equal(r.accepted, ref["accepted"]):
                        bad += 1; print("MISMATCH vs oracle", N, F, M, sd, edx)
                elif not (np.array_equal(r.dx, first.dx) and np.array_equal(r.accepted, first.accepted) and np.array_equal(r.P_new, first.P_new)):
                    bad += 1; print("NOT REPEATABLE", N, F, M, sd, it, np.abs(r.dx - first.dx).max(), (r.accepted != first.accepted).sum())
                    break
print("stress done, bad =", bad)
