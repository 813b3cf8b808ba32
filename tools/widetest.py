import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
with UpdateEngine(max_clones=31, max_features=2000, max_track=31) as e:
    for (N, F, M, kw, seed) in [(30, 60, 30, dict(variable_tracks=True, min_track=2), 1), (30, 40, 30, dict(variable_tracks=True, min_track=16), 2),
                                (20, 80, 20, dict(variable_tracks=True, min_track=2), 3), (31, 64, 31, {}, 24), (16, 50, 16, {}, 5),
                                (30, 300, 30, dict(variable_tracks=True, min_track=2, outlier_fraction=0.1, outlier_px=300.0), 6)]:
        prob = synth.make_problem(N, F, M, seed=seed, **kw)
        ref = oracle.update(prob, dense_noise=False)
        r = e.update_problem(prob)
        print(N, F, M, kw, "status", r.status, ref["status"], "mask", np.array_equal(r.accepted, ref["accepted"]), "dx %.2e P %.2e" % (rel(r.dx, ref["dx"]), rel(r.P_new, ref["P_new"])), "sym", np.array_equal(r.P_new, r.P_new.T), flush=True)
