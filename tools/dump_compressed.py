#!/usr/bin/env python3
"""Diagnostic: the compressed system (T, rn) of one small problem, written to an .npz; run under two MSCKF_LIB builds and compare."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
prob = synth.make_problem(N, F, M, seed=3)
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
    r = e.update_problem(prob)
    T, rn = e.debug_compressed()
np.savez(out, T=T, rn=rn, dx=r.dx)
if len(sys.argv) > 5:
    o = np.load(sys.argv[5])
    G1, G0 = T.T @ T, o["T"].T @ o["T"]
    print("|T^T T - ref| / |ref| = %.3e" % (np.linalg.norm(G1 - G0) / np.linalg.norm(G0)))
    np.set_printoptions(linewidth=250, precision=3)
    D = np.abs(np.abs(T) - np.abs(o["T"]))
    rows, cols = np.nonzero(D > 1e-9 * np.abs(o["T"]).max())
    print("differing entries (|T| vs |ref|):", len(rows), "rows", sorted(set(rows.tolist()))[:40], "cols", sorted(set(cols.tolist()))[:60])
    for rI in sorted(set(rows.tolist()))[:6]:
        print("row", rI, "new", T[rI, :40]); print("      ref", o["T"][rI, :40])
