#!/usr/bin/env python3
"""Where the host-inclusive call goes: wall time of update_problem, of the bare C call (arguments packed once),
and the device-side span the library reports (events around uploads + kernels)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd import synth, _ffi
from msckf_amd.api import UpdateEngine, chi2_table
N, F, M = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (30, 2000, 10))]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 300
prob = synth.make_problem(N, F, M, seed=0)
probs = [prob] + [synth.make_problem(N, F, M, seed=sd) for sd in (1, 2, 3)]
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
    for i in range(20):
        r = eng.update_problem(probs[i % 4])
    t0 = time.perf_counter()
    for i in range(iters):
        r = eng.update_problem(probs[i % 4])
    t_rot = (time.perf_counter() - t0) / iters * 1e6
    print(f"N={N} F={F} M={M}: update_problem over 4 rotating batches (plan cache misses) {t_rot:.0f} us = {1e6 / t_rot:.0f} updates/s")
with UpdateEngine(max_clones=N, max_features=F, max_track=M) as eng:
    for _ in range(20):
        r = eng.update_problem(prob)
    t0 = time.perf_counter()
    for _ in range(iters):
        r = eng.update_problem(prob)
    t_py = (time.perf_counter() - t0) / iters * 1e6
    a = eng._pack(prob); chi = _ffi.f64(chi2_table()); d = prob.d
    dx = np.empty(d); P_out = np.empty((d, d)); acc = np.zeros(F, dtype=np.uint8); st = _ffi.Stats()
    args = (eng._h, N, _ffi.dptr(a["P"]), _ffi.dptr(a["cam_R"]), _ffi.dptr(a["cam_t"]), _ffi.dptr(a["cam_R0"]), _ffi.dptr(a["cam_t0"]),
            _ffi.dptr(a["g"]), _ffi.dptr(a["Kinv"]), float(prob.sigma), F, _ffi.iptr(a["view_ptr"]), _ffi.dptr(a["obs_uv"]),
            _ffi.iptr(a["obs_slot"]), _ffi.dptr(a["idp_base"]), _ffi.dptr(a["idp_m"]), _ffi.dptr(a["idp_rho"]), _ffi.dptr(chi),
            int(chi.size), _ffi.dptr(dx), _ffi.dptr(P_out), _ffi.uptr(acc), C.byref(st))
    f = eng._lib.msckf_update
    t0 = time.perf_counter()
    for _ in range(iters):
        f(*args)
    t_c = (time.perf_counter() - t0) / iters * 1e6
    eng.load(prob); eng.run(); eng.sync()
    ms, _ = eng.run_timed(50)
    print(f"N={N} F={F} M={M}: update_problem {t_py:.0f} us  bare C call {t_c:.0f} us  device span in call {st.us_total:.0f} us  resident pipeline {ms / 50 * 1000:.0f} us")
    print(f"  -> python wrapper {t_py - t_c:.0f} us, C host side outside the device span {t_c - st.us_total:.0f} us, uploads etc inside the span {st.us_total - ms / 50 * 1000:.0f} us")
    print(f"  library stats: host_prep {st.us_host_prep:.0f} us, h2d {st.us_h2d:.0f} us, d2h {st.us_d2h:.0f} us")
    print(f"  host-inclusive {1e6 / t_py:.0f} updates/s (C-only caller: {1e6 / t_c:.0f}/s)")
