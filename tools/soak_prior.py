#!/usr/bin/env python3
"""Randomised parity soak with REALISTIC priors: P built the way a filter builds it -- propagation steps (P <- Phi P Phi^T + Q
on the IMU block and its cross terms) and clone augmentations (P <- [[P, P J^T], [J P, J P J^T]], reference MSCKF.py:236-265),
which leaves every clone block an exact linear image of the IMU block of its time: cond(P) ~ 1e12 - 1e18 with scales from
1e-8 (biases) to 1e-1 -- against recipe A's cond 1.4 --, ragged tracks (tools/soak_holes.py), outliers.
usage: soak_prior.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
from soak_holes import ragged


def realistic_prior(N, rng, steps_between=(1, 6)):
    P = np.diag(np.concatenate([np.full(3, 1e-4), np.full(3, 1e-7), np.full(3, 1e-2), np.full(3, 1e-5), np.full(3, 1e-2)]))
    for _ in range(N):
        for _ in range(int(rng.integers(*steps_between))):
            d = P.shape[0]
            Phi = np.eye(15) + 1e-2 * rng.standard_normal((15, 15)) * (rng.random((15, 15)) < 0.3)
            Q = np.diag(np.concatenate([np.full(3, 1e-7), np.full(3, 1e-10), np.full(3, 1e-5), np.full(3, 1e-9), np.full(3, 1e-7)]))
            Pn = P.copy()
            Pn[:15, :15] = Phi @ P[:15, :15] @ Phi.T + Q
            Pn[:15, 15:] = Phi @ P[:15, 15:]
            Pn[15:, :15] = Pn[:15, 15:].T
            P = (Pn + Pn.T) / 2
        d = P.shape[0]
        J = np.zeros((6, d))
        J[:3, :3] = np.eye(3) + 1e-2 * rng.standard_normal((3, 3))
        J[3:, 12:15] = np.eye(3)
        J[3:, :3] = 0.1 * rng.standard_normal((3, 3))                # lever arm
        M = np.vstack([np.eye(d), J])
        P = M @ P @ M.T
        P = (P + P.T) / 2
    return P


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    worst = 0.0
    bad = 0
    with UpdateEngine(max_clones=53, max_features=4096, max_track=31) as eng:
        for c in range(cases):
            N = int(rng.integers(2, 41)); F = int(rng.integers(1, 500))
            hi = int(rng.integers(2, min(N, 31) + 1))
            prob = ragged(rng, N, F, 2, hi, float(rng.choice([0.0, 0.1, 0.3])))
            prob.P = realistic_prior(N, rng)
            ref = oracle.update(prob, dense_noise=False)
            try:
                res = eng.update_problem(prob)
            except Exception as ex:
                bad += 1
                print(f"case {c}: N={N} F={F} views<={hi} cond {np.linalg.cond(prob.P):.1e}: {ex} | oracle status {ref['status']}", flush=True)
                continue
            gam, _ = eng.debug_gate()
            ok = res.status == ref["status"] and np.array_equal(res.accepted, ref["accepted"])
            e = 0.0
            if ok and res.status == 0:
                e = max(np.linalg.norm(res.dx - ref["dx"]) / max(np.linalg.norm(ref["dx"]), 1e-300),
                        np.linalg.norm(res.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"]))
            g = float(np.max(np.abs(gam - ref["gamma"]) / np.maximum(np.abs(ref["gamma"]), 1e-9)))
            worst = max(worst, e)
            if not ok or e > 1e-8 or g > 1e-6:
                bad += 1
                nd = int((res.accepted != ref["accepted"]).sum())
                print(f"case {c}: N={N} F={F} views<={hi} cond {np.linalg.cond(prob.P):.1e}: status {res.status}/{ref['status']} mask differs at {nd} err {e:.2e} gamma {g:.2e}", flush=True)
    print(f"{cases} cases, {bad} flagged, worst dx / P+ error {worst:.2e}")


if __name__ == "__main__":
    main()
