#!/usr/bin/env python3
"""Debug: wide sweep vs tree plan on ragged tracks: where does T^T T differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
np.set_printoptions(linewidth=250, precision=2)
N, F, M, seed = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (30, 500, 15, 38)
prob = synth.make_problem(N, F, M, seed=seed, variable_tracks=True)
vp = prob.view_ptr
lo = np.minimum.reduceat(prob.obs_slot, vp[:-1]); hi = np.maximum.reduceat(prob.obs_slot, vp[:-1])
print("spans:", np.bincount(hi - lo + 1))
with UpdateEngine(max_clones=N, max_features=F, max_track=M, plan="tree") as e:
    r0 = e.update_problem(prob); T0, z0 = e.debug_compressed()
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
    r1 = e.update_problem(prob); T1, z1 = e.debug_compressed()
    print("levels", r1.stats["n_levels"], "leaves", r1.stats["n_leaves"])
G0, G1 = T0.T @ T0, T1.T @ T1
err = np.abs(G0 - G1) / np.abs(G0).max()
print("max rel err G", err.max(), "dx err", np.linalg.norm(r0.dx - r1.dx) / np.linalg.norm(r0.dx))
bad = np.argwhere(err > 1e-10)
if len(bad):
    print("bad entries:", len(bad), "rows", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max())
    # first differing row of T (rows of R are unique up to sign)
    for i in range(6 * N):
        s = np.sign(T0[i, i]) * np.sign(T1[i, i]) if T0[i, i] != 0 and T1[i, i] != 0 else 1.0
        d = np.abs(T0[i] - s * T1[i]).max()
        if d > 1e-9 * np.abs(T0).max():
            print("first differing T row", i, "diff", d, "slot", i // 6)
            print("T0 row nz cols", np.nonzero(T0[i])[0][[0, -1]], "T1 row nz cols", (np.nonzero(T1[i])[0][[0, -1]] if T1[i].any() else None))
            break
    b0, b1 = T0.T @ z0, T1.T @ z1
    print("b err", np.abs(b0 - b1).max() / np.abs(b0).max())
