#!/usr/bin/env python3
"""A/B of host-side switches of the drop-in call in ONE session on ONE box (the host part varies from box to box):
usage: host_ab.py ENV_NAME  -> runs the rotating-batch loop in child processes with ENV_NAME=1 and =0 alternately."""
import os, subprocess, sys
name = sys.argv[1]
code = r'''
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
probs = [synth.make_problem(30, 2000, 10, seed=sd) for sd in range(4)]
with UpdateEngine(max_clones=30, max_features=2000, max_track=10) as eng:
    for i in range(40): eng.update_problem(probs[i % 4])
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(200): eng.update_problem(probs[i % 4])
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
print("%.1f" % best)
'''
for trial in range(3):
    for val in ("1", "0"):
        env = dict(os.environ); env[name] = val
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"{name}={val}: {out.stdout.strip()} us per call (best of 5 x 200)", flush=True)
