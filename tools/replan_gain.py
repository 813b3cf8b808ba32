#!/usr/bin/env python3
"""Fused select -> update with and without msckf_replan, at several valid fractions (device us, HIP events)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine

N, F, M = 30, 300, 10
prob = synth.make_problem(N, F, M, seed=3)
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
    for lost in (1.0, 0.5, 0.2, 0.1, 0.03):
        tracks = synth.make_tracks(prob, 3, lost_fraction=lost)
        sel = eng.select_problem(prob, tracks, synth.SelectParams(use_parallax=False, min_frames_tracked=2))
        eng.run(); eng.sync()
        ms_a, _ = eng.run_timed(30)
        lv_a = eng.result().stats["n_levels"]
        t0 = time.perf_counter(); eng.replan(); us_plan = (time.perf_counter() - t0) * 1e6
        eng.run(); eng.sync()
        ms_b, _ = eng.run_timed(30)
        lv_b = eng.result().stats["n_levels"]
        print(f"candidates {F}, valid {int(sel.valid.sum()):4d}: masked {ms_a / 30 * 1e3:6.0f} us ({lv_a} levels)   "
              f"replanned {ms_b / 30 * 1e3:6.0f} us ({lv_b} levels) + replan {us_plan:4.0f} us host", flush=True)
