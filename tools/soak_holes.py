#!/usr/bin/env python3
"""Randomised parity soak with RAGGED tracks: every track keeps a random subset of the views of a full-window track (holes,
any span, 2 - 31 views), random window sizes, outliers; the one-shot call against the oracle (1e-8, equal masks).
usage: soak_holes.py [cases] [seed] [f64|f32] [Fmin Fmax]   (f32: fp32 stack + f32 matrix-core products, tolerance 1e-4 / gamma 1e-3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle


def ragged(rng, N, F, keep_lo, keep_hi, outliers):
    full = synth.make_problem(N, F, N, seed=int(rng.integers(1 << 30)), outlier_fraction=outliers, outlier_px=300.0)
    vp = [0]; uv = []; sl = []
    for f in range(F):
        a, b = full.view_ptr[f], full.view_ptr[f + 1]
        k = int(rng.integers(keep_lo, min(keep_hi, b - a, 31) + 1))
        mode = rng.integers(3)
        if mode == 0:                                   # consecutive run
            s = int(rng.integers(0, b - a - k + 1)); idx = np.arange(s, s + k)
        elif mode == 1:                                 # random subset of the whole window
            idx = np.sort(rng.choice(b - a, size=k, replace=False))
        else:                                           # random subset of a window of at most 15 slots
            w = min(b - a, max(k, int(rng.integers(k, max(k, 15) + 1)))); s = int(rng.integers(0, b - a - w + 1))
            idx = s + np.sort(rng.choice(w, size=k, replace=False))
        uv.append(full.obs_uv[a + idx]); sl.append(full.obs_slot[a + idx]); vp.append(vp[-1] + k)
    q = synth.UpdateProblem(**{**full.__dict__})
    q.view_ptr = np.asarray(vp, dtype=np.int32); q.obs_uv = np.concatenate(uv); q.obs_slot = np.concatenate(sl).astype(np.int32)
    return q


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    worst = 0.0
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
    tol, gtol = (1e-8, 1e-7) if dtype == "f64" else (1e-4, 1e-3)
    Fmin, Fmax = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1, 400)
    # (SOAK_PLAN=band|tree: every track on the Householder plans / on the merge tree; the MSCKF_* switches select the fallbacks)
    with UpdateEngine(max_clones=53, max_features=max(2048, Fmax), max_track=31, dtype=dtype, plan=os.environ.get("SOAK_PLAN", "auto")) as eng:
        for c in range(cases):
            N = int(rng.integers(2, 54)); F = int(rng.integers(Fmin, Fmax))
            lo = 2; hi = int(rng.integers(2, min(N, 31) + 1))
            prob = ragged(rng, N, F, lo, hi, float(rng.choice([0.0, 0.1, 0.4])))
            ref = oracle.update(prob, dense_noise=False)
            try:
                res = eng.update_problem(prob)
            except Exception as ex:                     # (an engine error where the oracle has a result: keep the batch)
                print(f"case {c}: N={N} F={F} views<= {hi}: {ex} | oracle status {ref['status']} accepted {int(ref['accepted'].sum())}", flush=True)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", f"soak_fail_{c}.npz"), **{k: v for k, v in prob.__dict__.items() if k != "meta"})
                continue
            gam, _ = eng.debug_gate()
            ok = res.status == ref["status"] and np.array_equal(res.accepted, ref["accepted"])
            e = 0.0
            if ok and res.status == 0:
                e = max(np.linalg.norm(res.dx - ref["dx"]) / max(np.linalg.norm(ref["dx"]), 1e-300),
                        np.linalg.norm(res.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"]))
            gr = np.abs(gam - ref["gamma"]) / np.maximum(np.abs(ref["gamma"]), 1e-12)
            g = float(np.max(gr))
            kg = int(np.argmax(gr))
            worst = max(worst, e)
            if not ok or e > tol or g > gtol:
                print(f"case {c}: N={N} F={F} views<= {hi}: status {res.status}/{ref['status']} masks equal {np.array_equal(res.accepted, ref['accepted'])} err {e:.2e} gamma {g:.2e}"
                      f" (track {kg}: {int(prob.view_ptr[kg + 1] - prob.view_ptr[kg])} views, slots {prob.obs_slot[prob.view_ptr[kg]]}..{prob.obs_slot[prob.view_ptr[kg + 1] - 1]}, gamma {gam[kg]:.6e} / oracle {ref['gamma'][kg]:.6e})", flush=True)
    print(f"{cases} cases, worst dx / P+ error {worst:.2e}")


if __name__ == "__main__":
    main()
