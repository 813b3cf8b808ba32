#!/usr/bin/env python3
"""One named batch of tools/span_rows.py run a few times on the resident path (for rocprofv3 --kernel-trace: tools/ktimeline2.sh
u15|u30|frame|mixed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
which = sys.argv[1]
if which == "u15":
    prob = synth.make_problem(30, 2000, 15, seed=0, variable_tracks=True, min_track=2)
elif which == "u30":
    prob = synth.make_problem(30, 2000, 30, seed=0, variable_tracks=True, min_track=2)
elif which == "frame":      # a frame of the reference's size (main.py:199): 300 tracks ~ U[2, 30]
    prob = synth.make_problem(30, 300, 30, seed=0, variable_tracks=True, min_track=2)
else:
    a = synth.make_problem(30, 1990, 10, seed=0)
    b = synth.make_problem(30, 10, 30, seed=100, P=a.P, poses=(a.cam_R, a.cam_t))
    vp = np.concatenate([a.view_ptr, a.view_ptr[-1] + b.view_ptr[1:]])
    cat = lambda x, y: np.concatenate([x, y])
    prob = synth.UpdateProblem(**{**a.__dict__, "view_ptr": vp, "obs_uv": cat(a.obs_uv, b.obs_uv), "obs_slot": cat(a.obs_slot, b.obs_slot),
                                  "idp_base": cat(a.idp_base, b.idp_base), "idp_m": cat(a.idp_m, b.idp_m), "idp_rho": cat(a.idp_rho, b.idp_rho)})
with UpdateEngine(max_clones=30, max_features=2000, max_track=30) as eng:
    eng.load(prob)
    for _ in range(3):
        eng.run()
    ms, st = eng.run_timed(3, stages=True)
    print(which, f"{ms / 3 * 1000:.0f} us/update", st)
