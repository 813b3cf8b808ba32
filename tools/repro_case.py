#!/usr/bin/env python3
"""One saved batch (npz of an UpdateProblem: tools/soak_holes.py's failures, gpurun_out/soak_fail_*.npz) through the one-shot call
against the oracle: dx / P+ error, the tracks whose gate statistic differs most.  usage: repro_case.py file.npz [f64|f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
z = np.load(sys.argv[1])
prob = synth.UpdateProblem(**{k: (z[k] if z[k].shape else z[k].item()) for k in z.files})
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
ref = oracle.update(prob, dense_noise=False)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
with UpdateEngine(max_clones=53, max_features=max(2048, prob.F), max_track=31, dtype=dtype) as eng:
    res = eng.update_problem(prob)
    gam, q = eng.debug_gate()
    print("status", res.status, ref["status"], "masks equal", np.array_equal(res.accepted, ref["accepted"]), "dx", rel(res.dx, ref["dx"]), "P", rel(res.P_new, ref["P_new"]))
    print("split", eng.debug_split())
    gr = np.abs(gam - ref["gamma"]) / np.maximum(np.abs(ref["gamma"]), 1e-12)
    for k in np.argsort(-gr)[:8]:
        sl = prob.obs_slot[prob.view_ptr[k]:prob.view_ptr[k + 1]]
        print(f"  track {k}: {len(sl)} views slots {sl.min()}..{sl.max()} span {sl.max() - sl.min() + 1} ordered {bool(np.all(np.diff(sl) > 0))} gamma rel diff {gr[k]:.2e} ({gam[k]:.6e} / {ref['gamma'][k]:.6e}) accepted {int(ref['accepted'][k])}")
