import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
worst = 0.0
n = 0
with UpdateEngine(max_clones=53, max_features=600, max_track=20) as e:
    for seed in range(20, 28):
        rng = np.random.default_rng(seed)
        for _ in range(40):
            N = int(rng.integers(2, 54)); M = int(rng.integers(2, min(N, 20) + 1)); F = int(rng.integers(1, 600))
            kw = {}
            if rng.random() < 0.5: kw["variable_tracks"] = True
            if rng.random() < 0.3: kw.update(outlier_fraction=0.1, outlier_px=300.0)
            sd = int(rng.integers(0, 10 ** 6))
            prob = synth.make_problem(N, F, M, seed=sd, **kw)
            ref = oracle.update(prob, dense_noise=False)
            r = e.update_problem(prob)
            assert r.status == ref["status"], (N, F, M, sd, kw)
            assert np.array_equal(r.accepted, ref["accepted"]), (N, F, M, sd, kw)
            if ref["status"] == 0:
                edx = np.linalg.norm(r.dx - ref["dx"]) / np.linalg.norm(ref["dx"]); eP = np.linalg.norm(r.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"])
                assert edx < 1e-8 and eP < 1e-8, (N, F, M, sd, kw, edx, eP)
                worst = max(worst, edx, eP)
            n += 1
        print("seed", seed, "ok, worst so far %.2e" % worst, flush=True)
print("SOAK OK", n, "cases, worst %.2e" % worst)
