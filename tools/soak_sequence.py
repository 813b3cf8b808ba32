#!/usr/bin/env python3
"""Randomised soak of the RESIDENT calls in filter order (reference MSCKF.py:236-265 process_imu / state_augmentation, :570-614
update + correct, :751-757 remove_cameras): random interleavings of msckf_propagate, msckf_augment, msckf_remove_clones, resident
updates on ragged batches (set_features -> run -> get_result(dx) -> commit_covariance -> set_poses, none of them waited for) and
one-shot updates in between, the covariance compared with the oracle's after every step.
usage: soak_sequence.py [sequences] [seed] [steps per sequence]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def ragged_on(full, rng, hi):
    """`full` (every track sees every clone) thinned to random subsets of 2..hi views per track."""
    vp = [0]; uv = []; sl = []
    for f in range(full.F):
        a, b = full.view_ptr[f], full.view_ptr[f + 1]
        k = int(rng.integers(2, min(hi, b - a, 31) + 1))
        if rng.integers(2):
            s = int(rng.integers(0, b - a - k + 1)); idx = np.arange(s, s + k)
        else:
            idx = np.sort(rng.choice(b - a, size=k, replace=False))
        uv.append(full.obs_uv[a + idx]); sl.append(full.obs_slot[a + idx]); vp.append(vp[-1] + k)
    q = synth.UpdateProblem(**{**full.__dict__})
    q.view_ptr = np.asarray(vp, dtype=np.int32); q.obs_uv = np.concatenate(uv); q.obs_slot = np.concatenate(sl).astype(np.int32)
    return q


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 29)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    worst, bad = 0.0, 0
    MAXN = 24
    with UpdateEngine(max_clones=MAXN, max_features=1024, max_track=MAXN) as eng:
        for s in range(nseq):
            N0 = int(rng.integers(2, 8))
            st = synth.make_problem(N0, 4, 2, seed=int(rng.integers(1 << 30)))
            P, cam_R, cam_t = st.P.copy(), st.cam_R.copy(), st.cam_t.copy()
            eng.set_prior(P, st.gravity, st.K, st.sigma, cam_R, cam_t)
            log = []
            for k in range(steps):
                N = cam_R.shape[0]
                op = rng.choice(["prop", "aug", "rem", "upd", "upd", "oneshot"])
                if op == "prop":
                    Phi = np.eye(15) + 0.01 * rng.standard_normal((15, 15)); A = rng.standard_normal((15, 15)); Q = 1e-6 * A @ A.T
                    eng.propagate(Phi, Q); P = oracle.propagate_covariance(P, Phi, Q)
                elif op == "aug" and N < MAXN:
                    J = np.zeros((6, 15)); J[:3, :3] = np.eye(3) + 0.01 * rng.standard_normal((3, 3)); J[3:, 12:] = np.eye(3)
                    J[3:, :3] = 0.05 * rng.standard_normal((3, 3))
                    nR, nt = cam_R[-1], cam_t[-1] + np.array([0.15, 0.0, 0.0])
                    eng.augment(J, nR, nt); P = oracle.augment_covariance(P, J)
                    cam_R = np.concatenate([cam_R, nR[None]]); cam_t = np.concatenate([cam_t, nt[None]])
                elif op == "rem" and N > 3:
                    drop = sorted(set(int(x) for x in rng.choice(N, size=int(rng.integers(1, 3)), replace=False)))
                    eng.remove_clones(drop); P = oracle.remove_clones_covariance(P, drop)
                    keep = [i for i in range(N) if i not in drop]
                    cam_R, cam_t = cam_R[keep], cam_t[keep]
                elif op in ("upd", "oneshot") and N >= 2:
                    F = int(rng.integers(1, 300))
                    full = synth.make_problem(N, F, N, seed=int(rng.integers(1 << 30)), P=P, poses=(cam_R, cam_t),
                                              outlier_fraction=float(rng.choice([0.0, 0.2])), outlier_px=300.0)
                    batch = ragged_on(full, rng, int(rng.integers(2, N + 1)))
                    exp = oracle.update(batch)
                    if op == "oneshot":                                   # must not touch the resident covariance
                        res = eng.update_problem(batch)
                        ok = res.status == exp["status"] and np.array_equal(res.accepted, exp["accepted"])
                        e = max(rel(res.dx, exp["dx"]), rel(res.P_new, exp["P_new"])) if ok and res.status == 0 else 0.0
                        if not ok or e > 1e-8:
                            bad += 1; print(f"seq {s} step {k}: one-shot update N={N} F={F}: status {res.status}/{exp['status']} err {e:.2e}", flush=True)
                        # (the one-shot call loaded its own state: put the resident one back, as a caller mixing the two would)
                        eng.set_prior(P, st.gravity, st.K, st.sigma, cam_R, cam_t)
                    else:
                        eng.set_features(batch); eng.run()
                        res = eng.result()
                        ok = res.status == exp["status"] and np.array_equal(res.accepted, exp["accepted"])
                        if ok and res.status == 0:
                            assert eng.commit_covariance() == 0
                            P = exp["P_new"]
                            cam_t = cam_t + 1e-4 * rng.standard_normal(cam_t.shape)        # "state injection"
                            eng.set_poses(cam_R, cam_t)
                        if not ok:
                            bad += 1; print(f"seq {s} step {k}: resident update N={N} F={F}: status {res.status}/{exp['status']} mask {np.array_equal(res.accepted, exp['accepted'])}", flush=True)
                else:
                    continue
                log.append(op)
                e = rel(eng.covariance(), P)
                worst = max(worst, e)
                if e > 1e-9 or eng.n_clones != cam_R.shape[0]:
                    bad += 1
                    print(f"seq {s} step {k} after {op}: covariance err {e:.2e}, clones {eng.n_clones}/{cam_R.shape[0]} (ops so far: {' '.join(log[-8:])})", flush=True)
                    break
    print(f"{nseq} sequences x {steps} steps, {bad} flagged, worst covariance error {worst:.2e}")


if __name__ == "__main__":
    main()
