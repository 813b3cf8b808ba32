#!/usr/bin/env python3
"""Resident device time of the mixed-span batches (the rows VERDICT r3 asked for): tracks ~ U[2, 30], ~ U[2, 15], 1990 ten-view +
10 thirty-view tracks, against the pure ten-view batch; parity of each against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
check = "--check" in sys.argv


def mixed(seed=0):
    a = synth.make_problem(30, 1990, 10, seed=seed)
    b = synth.make_problem(30, 10, 30, seed=seed + 100, P=a.P, poses=(a.cam_R, a.cam_t))
    vp = np.concatenate([a.view_ptr, a.view_ptr[-1] + b.view_ptr[1:]])
    cat = lambda x, y: np.concatenate([x, y])
    return synth.UpdateProblem(**{**a.__dict__, "view_ptr": vp, "obs_uv": cat(a.obs_uv, b.obs_uv), "obs_slot": cat(a.obs_slot, b.obs_slot),
                                  "idp_base": cat(a.idp_base, b.idp_base), "idp_m": cat(a.idp_m, b.idp_m), "idp_rho": cat(a.idp_rho, b.idp_rho)})


rows = [("pure ten-view", synth.make_problem(30, 2000, 10, seed=0)),
        ("1990 ten-view + 10 thirty-view", mixed()),
        ("track ~ U[2, 15]", synth.make_problem(30, 2000, 15, seed=0, variable_tracks=True, min_track=2)),
        ("track ~ U[2, 30]", synth.make_problem(30, 2000, 30, seed=0, variable_tracks=True, min_track=2))]
with UpdateEngine(max_clones=30, max_features=2000, max_track=30) as eng:
    for name, prob in rows:
        eng.load(prob)
        for _ in range(3):
            eng.run()
        ms, st = eng.run_timed(30, stages=True)
        line = f"{name:34s} {ms / 30 * 1000:7.1f} us/update  stages {[round(x, 1) for x in st]}"
        if check:
            from oracle import msckf_oracle as oracle
            ref = oracle.update(prob, dense_noise=False)
            r = eng.update_problem(prob)
            rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
            line += f"  mask {np.array_equal(r.accepted, ref['accepted'])} dx {rel(r.dx, ref['dx']):.1e} P {rel(r.P_new, ref['P_new']):.1e}"
        print(line, flush=True)
