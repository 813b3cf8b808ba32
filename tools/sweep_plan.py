#!/usr/bin/env python3
"""Sweep QR-tree plan parameters (leaf rows, merge arity) at one problem size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
N, F, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (30, 2000, 10)))
prob = synth.make_problem(N, F, M, seed=0)
ref = None
for leaf_rows in (96, 128, 160, 192, 256):
    for arity in (2, 3, 4, 6, 8):
        with UpdateEngine(max_clones=N, max_features=F, max_track=M, leaf_rows=leaf_rows, merge_arity=arity) as eng:
            eng.load(prob)
            for _ in range(3):
                eng.run()
            ms, st = eng.run_timed(20, stages=True)
            res = eng.result()
            if ref is None:
                ref = res
            err = np.linalg.norm(res.dx - ref.dx) / np.linalg.norm(ref.dx)
            print(f"leaf_rows={leaf_rows:4d} arity={arity}  leaves={res.stats['n_leaves']:4d} levels={res.stats['n_levels']}  "
                  f"total={ms/20*1000:7.1f} us  qr={st[1]:7.1f} us  dx_dev={err:.1e}", flush=True)
