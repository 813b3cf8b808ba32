"""The drop-in boundary from plain C: tests/c_abi/drop_in.c is built with gcc against include/msckf_mi355x.h and
libmsckf_mi355x.so, fed one update problem through a flat file and compared with the oracle (1e-8 on dx and P+,
mask bit-equal) -- the C-ABI needs neither Python nor PyTorch nor HIP headers on the caller's side."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu


def _write_problem(path, prob, chi2):
    Kinv = np.linalg.inv(np.asarray(prob.K, dtype=np.float64))
    with open(path, "wb") as f:
        np.array([prob.N, prob.F, int(prob.view_ptr[-1]), chi2.size], dtype=np.int32).tofile(f)
        for a in (prob.P, prob.cam_R, prob.cam_t, prob.cam_R0, prob.cam_t0, prob.gravity, Kinv, np.array([prob.sigma]),
                  prob.obs_uv, prob.idp_base, prob.idp_m, prob.idp_rho, chi2):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
        np.ascontiguousarray(prob.view_ptr, dtype=np.int32).tofile(f)
        np.ascontiguousarray(prob.obs_slot, dtype=np.int32).tofile(f)


@pytest.mark.parametrize("N,F,M,kw", [(10, 50, 5, {"outlier_fraction": 0.1, "outlier_px": 500.0}), (30, 2000, 10, {})])
def test_c_caller_matches_oracle(tmp_path, N, F, M, kw):
    from msckf_amd import synth
    from msckf_amd.api import chi2_table
    from oracle import msckf_oracle as oracle
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    pkg = os.path.join(ROOT, "monocular-visual-inertial-msckf_amd")
    exe = str(tmp_path / "drop_in")
    subprocess.run(["gcc", "-O2", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "drop_in.c"),
                    "-L", pkg, "-lmsckf_mi355x", "-Wl,-rpath," + pkg, "-o", exe], check=True)
    prob = synth.make_problem(N, F, M, seed=3, **kw)
    ref = oracle.update(prob, dense_noise=False)
    pin, pout = str(tmp_path / "p.bin"), str(tmp_path / "r.bin")
    _write_problem(pin, prob, chi2_table())
    r = subprocess.run([exe, pin, pout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = prob.d
    with open(pout, "rb") as f:
        status = int(np.fromfile(f, dtype=np.int32, count=1)[0])
        dx = np.fromfile(f, dtype=np.float64, count=d)
        P = np.fromfile(f, dtype=np.float64, count=d * d).reshape(d, d)
        acc = np.fromfile(f, dtype=np.uint8, count=prob.F)
    assert status == ref["status"] == 0
    assert np.array_equal(acc, ref["accepted"].astype(np.uint8))
    assert rel_err(dx, ref["dx"]) < 1e-8 and rel_err(P, ref["P_new"]) < 1e-8
