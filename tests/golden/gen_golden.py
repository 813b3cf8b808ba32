#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE.

Runs only in the build container, where `/root/reference` is mounted; it never
runs on the GPU box and nothing here is imported by tests.  It imports the
reference's own `MSCKF` class (stub modules stand in for the cv2 / rerun /
IPython / XFeat imports its module headers pull in but `update`/`correct`
never touch -- SURVEY.md Appendix A), builds the reference's objects from a
seeded synthetic `UpdateProblem`, calls the reference's `MSCKF.update(features)`
and stores the flat inputs and the captured outputs as `<case>.npz`.

Only numeric arrays are written -- no reference source or bytecode.

    python tests/golden/gen_golden.py            # all cases except the 45 s headline
    python tests/golden/gen_golden.py --headline # also cfg3 (needs ~12 GB RAM, ~1 min)
"""
import argparse
import copy
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

for name in ["cv2", "rerun", "IPython", "IPython.display", "modules", "modules.xfeat"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["cv2"].Mat = object
sys.modules["cv2"].addWeighted = lambda a, wa, b, wb, g: a      # debug overlay of add_camera_measurements (MSCKF.py:336), not arithmetic
sys.modules["IPython.display"].display = lambda *a, **k: None
sys.modules["IPython.display"].clear_output = lambda *a, **k: None


class _XFeat:
    def __init__(self, *a, **k):
        pass


sys.modules["modules.xfeat"].XFeat = _XFeat
sys.path.insert(0, "/root/reference")

from src.msckf.MSCKF import MSCKF, MSCKFParameters  # noqa: E402
from src.msckf.Camera import Camera  # noqa: E402
from src.msckf.FeatureExtractor import Feature  # noqa: E402
from src.msckf.IMU import IMUMeasurement  # noqa: E402
from src.utils.geometry import Isometry3D, InverseDepthPoint, Line  # noqa: E402

import scipy  # noqa: E402
from scipy.stats import chi2  # noqa: E402

import msckf_amd  # noqa: E402
from msckf_amd import synth  # noqa: E402


def realistic_state(N, seed):
    """Recipe B: drive the reference's own process_imu x10 + state_augmentation
    per clone from a 15x15 diagonal prior.  Returns (P, cam_R, cam_t, keys)."""
    rng = np.random.default_rng(1000 + seed)
    f = MSCKF(MSCKFParameters())
    f.state.imu.is_initialized = True
    f.first_measurement_arrived = True
    f.state.imu.v_W_Ii = np.array([1.2, 0.0, 0.0])
    f.state.imu.v_W_Ii_null = np.array([1.2, 0.0, 0.0])
    f.state.imu.gyroscope_bias = np.zeros(3)
    f.state.imu.accelerometer_bias = np.zeros(3)
    f.state.covariance = np.diag([1e-4] * 3 + [1e-6] * 3 + [1e-3] * 3 + [1e-5] * 3 + [1e-3] * 3).astype(float)
    t = 0.0
    g = f.state.imu.W_gravity
    for c in range(N):
        for k in range(10):
            t += 0.01
            R = f.state.imu.T_W_Ii.R
            a_w = np.array([0.3, 0.8 * np.cos(2.0 * t), 0.3 * np.sin(1.3 * t)])
            acc = R.T @ (a_w + g) + 1e-3 * rng.standard_normal(3)
            gyro = np.array([0.02 * np.sin(t), 0.03 * np.cos(0.7 * t), 0.05 * np.sin(0.5 * t)]) + 1e-4 * rng.standard_normal(3)
            f.process_imu(IMUMeasurement(t, gyro, acc))
        f.state_augmentation()
    keys = list(f.state.cameras.keys())
    cam_R = np.stack([f.state.cameras[k].T_W_Ci.R for k in keys])
    cam_t = np.stack([f.state.cameras[k].T_W_Ci.t for k in keys])
    return f.state.covariance.copy(), cam_R, cam_t, keys


def run_reference(prob, keys=None, imu_seed=0):
    """Build reference objects from the flat problem, call MSCKF.update, capture."""
    N, F = prob.N, prob.F
    params = MSCKFParameters()
    params.K = prob.K
    params.sigma_image = prob.sigma
    params.W_gravity = prob.gravity.copy()
    f = MSCKF(params)
    rng = np.random.default_rng(7 + imu_seed)
    imu_R = synth.so3_exp(0.1 * rng.standard_normal(3))
    f.state.imu.T_W_Ii = Isometry3D(imu_R.copy(), rng.standard_normal(3))
    f.state.imu.v_W_Ii = rng.standard_normal(3)
    f.state.imu.gyroscope_bias = 1e-3 * rng.standard_normal(3)
    f.state.imu.accelerometer_bias = 1e-2 * rng.standard_normal(3)
    imu0 = dict(imu_R=f.state.imu.T_W_Ii.R.copy(), imu_t=f.state.imu.T_W_Ii.t.copy(),
                imu_v=f.state.imu.v_W_Ii.copy(), imu_bg=f.state.imu.gyroscope_bias.copy(),
                imu_ba=f.state.imu.accelerometer_bias.copy())
    if keys is None:
        keys = [10 * (i + 1) for i in range(N)]
    for i, k in enumerate(keys):
        cam = Camera(prob.K, 640, 480, Isometry3D(prob.cam_R[i].copy(), prob.cam_t[i].copy()))
        if not (np.array_equal(prob.cam_R0[i], prob.cam_R[i]) and np.array_equal(prob.cam_t0[i], prob.cam_t[i])):
            cam.T_W_Ci_null = Isometry3D(prob.cam_R0[i].copy(), prob.cam_t0[i].copy())
        f.state.cameras[k] = cam
    f.state.covariance = prob.P.copy()
    feats = {}
    for j in range(F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        ft = Feature()
        ft.keypoints = [prob.obs_uv[i].copy() for i in range(a, b)]
        ft.camera_indices = [keys[int(prob.obs_slot[i])] for i in range(a, b)]
        idp = InverseDepthPoint()
        idp.base = prob.idp_base[j].copy()
        idp.m = prob.idp_m[j].copy()
        idp.rho = float(prob.idp_rho[j])
        ft.inverse_depth_point = idp
        feats[100 + j] = ft

    # per-feature gate statistics through the reference's own jacobian code
    gamma = np.zeros(F)
    crit = np.zeros(F)
    accepted = np.zeros(F, dtype=np.uint8)
    d = prob.d
    G = np.zeros((d, d))
    bvec = np.zeros(d)
    for j, ft in enumerate(feats.values()):
        r_o, H_o = f.compute_residual_and_jacobians(ft)
        S = H_o @ f.state.covariance @ H_o.T + f.sigma_image ** 2 * np.eye(H_o.shape[0])
        gamma[j] = float((r_o.T @ np.linalg.inv(S) @ r_o).flatten()[0])
        crit[j] = float(chi2.ppf(0.95, r_o.shape[0]))
        accepted[j] = 1 if f.gating_test(r_o, H_o) else 0
        if accepted[j]:
            G += H_o.T @ H_o
            bvec += (H_o.T @ r_o).flatten()

    captured = {}
    orig_correct = f.correct

    def wrapped(Kg, T_H, R_n, delta_x):
        captured["T_H"] = np.array(T_H)
        captured["dx"] = np.array(delta_x).flatten()
        captured["Rn_dev"] = float(np.abs(R_n - f.sigma_image ** 2 * np.eye(R_n.shape[0])).max())
        return orig_correct(Kg, T_H, R_n, delta_x)

    f.correct = wrapped
    rej0 = f.number_of_residuals_discarded_for_gasting_test
    f.update(feats)
    n_rej = f.number_of_residuals_discarded_for_gasting_test - rej0
    status = 0 if "dx" in captured else 1
    out = dict(
        status=np.int32(status),
        dx=captured.get("dx", np.zeros(d)),
        P_new=f.state.covariance.copy(),
        accepted=accepted, gamma=gamma, crit=crit, n_rejected=np.int32(n_rej),
        G=G, b=bvec,
        ThT_Th=(captured["T_H"].T @ captured["T_H"]) if status == 0 else np.zeros((d, d)),
        Rn_dev=np.float64(captured.get("Rn_dev", 0.0)),
        post_imu_R=f.state.imu.T_W_Ii.R.copy(), post_imu_t=f.state.imu.T_W_Ii.t.copy(),
        post_imu_v=f.state.imu.v_W_Ii.copy(), post_imu_bg=f.state.imu.gyroscope_bias.copy(),
        post_imu_ba=f.state.imu.accelerometer_bias.copy(),
        post_cam_R=np.stack([f.state.cameras[k].T_W_Ci.R for k in keys]),
        post_cam_t=np.stack([f.state.cameras[k].T_W_Ci.t for k in keys]),
        min_gate_margin=np.float64(np.min(np.abs(gamma - crit) / crit)) if F else np.float64(1.0),
        **imu0)
    assert int(accepted.sum()) + int(n_rej) == F
    return out


def run_reference_select(prob, tracks, sp, keys=None):
    """f1: build the reference's Feature objects (with lines and frame counters), call the
    reference's `get_valid_features`, then `update(valid_features)` as `process_features` does
    (`MSCKF.py:450-456`).  Returns the selection outputs and the chained update outputs."""
    N, F = prob.N, prob.F
    params = MSCKFParameters()
    params.K = prob.K
    params.sigma_image = prob.sigma
    params.W_gravity = prob.gravity.copy()
    params.width, params.height = sp.width, sp.height
    params.use_parallax = sp.use_parallax
    params.min_parallax = sp.min_parallax_deg
    params.min_number_of_frames_to_be_lost = sp.min_frames_lost
    params.min_number_of_frames_to_be_tracked = sp.min_frames_tracked
    f = MSCKF(params)
    if keys is None:
        keys = [10 * (i + 1) for i in range(N)]
    for i, k in enumerate(keys):
        f.state.cameras[k] = Camera(prob.K, sp.width, sp.height, Isometry3D(prob.cam_R[i].copy(), prob.cam_t[i].copy()))
    f.state.covariance = prob.P.copy()
    feats = {}
    for j in range(F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        ft = Feature()
        ft.keypoints = [prob.obs_uv[i].copy() for i in range(a, b)]
        ft.camera_indices = [keys[int(prob.obs_slot[i])] for i in range(a, b)]
        ft.lines = [Line(tracks.line_base[i].copy(), tracks.line_dir[i].copy(), float(tracks.line_conf[i]))
                    for i in range(a, b)]
        ft.lost_for_n_frames = int(tracks.lost_for[j])
        ft.tracked_for_n_frames = int(tracks.tracked_for[j])
        idp = InverseDepthPoint()
        idp.base = prob.idp_base[j].copy()
        idp.m = prob.idp_m[j].copy()
        idp.rho = float(prob.idp_rho[j])
        ft.inverse_depth_point = idp
        feats[100 + j] = ft
    f.estimated_world_points = []
    f.currently_processed_world_points = []
    valid, lost = f.get_valid_features(feats)
    flags = np.zeros(F, dtype=np.uint8)
    idp_m = np.stack([feats[100 + j].inverse_depth_point.m for j in range(F)])
    idp_rho = np.array([feats[100 + j].inverse_depth_point.rho for j in range(F)])
    for j in range(F):
        k = 100 + j
        refreshed = not (np.array_equal(idp_m[j], prob.idp_m[j]) and idp_rho[j] == prob.idp_rho[j])
        flags[j] = (1 if k in valid else 0) | (2 if k in lost else 0) | (4 if refreshed else 0)
    world = np.full((F, 3), np.nan)
    refreshed_idx = [j for j in range(F) if flags[j] & 4]
    assert len(refreshed_idx) == len(f.estimated_world_points)
    for j, w in zip(refreshed_idx, f.estimated_world_points):      # appended in dict order, MSCKF.py:489
        world[j] = w
    captured = {}
    orig_correct = f.correct

    def wrapped(Kg, T_H, R_n, delta_x):
        captured["dx"] = np.array(delta_x).flatten()
        return orig_correct(Kg, T_H, R_n, delta_x)

    f.correct = wrapped
    rej0 = f.number_of_residuals_discarded_for_gasting_test
    if len(valid) > 0:
        f.update(valid)
    d = prob.d
    return dict(sel_flags=flags, sel_idp_m=idp_m, sel_idp_rho=idp_rho, sel_world=world,
                status=np.int32(0 if "dx" in captured else 1), dx=captured.get("dx", np.zeros(d)),
                P_new=f.state.covariance.copy(),
                n_rejected=np.int32(f.number_of_residuals_discarded_for_gasting_test - rej0))


def save_select(name, prob, tracks, sp, out):
    arrays = dict(
        P=prob.P, cam_R=prob.cam_R, cam_t=prob.cam_t, cam_R0=prob.cam_R0, cam_t0=prob.cam_t0,
        gravity=prob.gravity, K=prob.K, sigma=np.float64(prob.sigma), view_ptr=prob.view_ptr,
        obs_uv=prob.obs_uv, obs_slot=prob.obs_slot, idp_base=prob.idp_base, idp_m=prob.idp_m,
        idp_rho=prob.idp_rho,
        line_base=tracks.line_base, line_dir=tracks.line_dir, line_conf=tracks.line_conf,
        lost_for=tracks.lost_for, tracked_for=tracks.tracked_for,
        select_params=np.array([sp.min_frames_lost, sp.min_frames_tracked, int(sp.use_parallax),
                                sp.min_parallax_deg, sp.width, sp.height], dtype=np.float64),
        versions=np.array([np.__version__, scipy.__version__, sys.version.split()[0]]))
    arrays.update(out)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    fl = out["sel_flags"]
    print(f"{name:28s} N={prob.N:3d} F={prob.F:5d} valid={int((fl & 1).sum()):4d} lost={int(((fl & 2) > 0).sum()):4d} "
          f"refreshed={int(((fl & 4) > 0).sum()):4d} status={int(out['status'])} rejected={int(out['n_rejected'])} "
          f"size={os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def run_reference_sequence(seed, frames=6, imu_per_frame=8, F=40, M=5, max_clones=6):
    """f2/f3: a short filter run through the reference's own `process_imu` (x imu_per_frame),
    `state_augmentation`, `update` and `remove_cameras` (`MSCKF.py:160-265, 570-661, 751-758`),
    recording what each step reads and the covariance it leaves.  Op kinds: 0 imu, 1 augment,
    2 update, 3 remove clones."""
    rng = np.random.default_rng(9000 + seed)
    params = MSCKFParameters()
    f = MSCKF(params)
    imu = f.state.imu
    imu.is_initialized = True
    f.first_measurement_arrived = True
    imu.v_W_Ii = np.array([1.0, 0.1, 0.0])
    imu.v_W_Ii_null = np.array([1.0, 0.1, 0.0])
    imu.gyroscope_bias = 1e-3 * rng.standard_normal(3)
    imu.accelerometer_bias = 1e-2 * rng.standard_normal(3)
    f.state.covariance = np.diag([1e-4] * 3 + [1e-6] * 3 + [1e-3] * 3 + [1e-5] * 3 + [1e-3] * 3).astype(float)
    arrays = dict(P0=f.state.covariance.copy(), Qc=f.continuous_noise_covariance.copy(),
                  T_W_I_R=imu.T_W_I.R.copy(), T_W_I_t=imu.T_W_I.t.copy(),
                  T_W_C_R=np.array(f.T_W_C.R, dtype=float), T_W_C_t=np.array(f.T_W_C.t, dtype=float),
                  gravity=np.array(imu.W_gravity, dtype=float), K=np.array(f.K), sigma=np.float64(f.sigma_image))
    kinds = []
    snap = {}
    orig_integrate = imu.integrate

    def integrate(linear_acceleration, angular_velocity, dt):
        orig_integrate(linear_acceleration=linear_acceleration, angular_velocity=angular_velocity, dt=dt)
        snap.update(acc=np.array(linear_acceleration), gyro=np.array(angular_velocity), dt=np.float64(dt),
                    R=imu.T_W_Ii.R.copy(), t=imu.T_W_Ii.t.copy(), v=imu.v_W_Ii.copy(),
                    R0=imu.T_W_Ii_null.R.copy(), t0=imu.T_W_Ii_null.t.copy(), v0=imu.v_W_Ii_null.copy(),
                    w_planet=np.array(imu.planet_angular_velocity, dtype=float))

    imu.integrate = integrate

    def put(kind, **kw):
        i = len(kinds)
        kinds.append(kind)
        for k, v in kw.items():
            arrays[f"o{i}_{k}"] = np.asarray(v)

    tnow = 0.0
    for frame in range(frames):
        for _ in range(imu_per_frame):
            tnow += 0.005
            R = imu.T_W_Ii.R
            a_w = np.array([0.4 * np.cos(1.5 * tnow), 0.6 * np.sin(2.0 * tnow), 0.2 * np.sin(1.1 * tnow)])
            acc = R.T @ (a_w + imu.W_gravity) + 1e-3 * rng.standard_normal(3)
            gyro = np.array([0.05 * np.sin(tnow), 0.04 * np.cos(0.7 * tnow), 0.1 * np.sin(0.5 * tnow)]) + 1e-4 * rng.standard_normal(3)
            f.process_imu(IMUMeasurement(tnow, gyro, acc))
            put(0, **{k: v for k, v in snap.items()}, P_after=f.state.covariance.copy())
        imu_R, imu_t = imu.T_W_Ii.R.copy(), imu.T_W_Ii.t.copy()
        f.state_augmentation()
        new_key = list(f.state.cameras.keys())[-1]
        cam = f.state.cameras[new_key]
        put(1, imu_R=imu_R, imu_t=imu_t, cam_R=cam.T_W_Ci.R.copy(), cam_t=cam.T_W_Ci.t.copy(),
            P_after=f.state.covariance.copy())
        keys = list(f.state.cameras.keys())
        N = len(keys)
        if N >= 3:
            cam_R = np.stack([f.state.cameras[k].T_W_Ci.R for k in keys])
            cam_t = np.stack([f.state.cameras[k].T_W_Ci.t for k in keys])
            prob = synth.make_problem(N, F, min(M, N), seed=100 * seed + frame, P=f.state.covariance.copy(),
                                      poses=(cam_R, cam_t), variable_tracks=True, min_track=2)
            feats = {}
            for j in range(prob.F):
                a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
                ft = Feature()
                ft.keypoints = [prob.obs_uv[i].copy() for i in range(a, b)]
                ft.camera_indices = [keys[int(prob.obs_slot[i])] for i in range(a, b)]
                idp = InverseDepthPoint()
                idp.base, idp.m, idp.rho = prob.idp_base[j].copy(), prob.idp_m[j].copy(), float(prob.idp_rho[j])
                ft.inverse_depth_point = idp
                feats[j] = ft
            pre = dict(imu_R=imu.T_W_Ii.R.copy(), imu_t=imu.T_W_Ii.t.copy(), imu_v=imu.v_W_Ii.copy(),
                       imu_bg=imu.gyroscope_bias.copy(), imu_ba=imu.accelerometer_bias.copy())
            cap = {}
            orig_correct = f.correct

            def wrapped(Kg, T_H, R_n, delta_x, _cap=cap, _oc=orig_correct):
                _cap["dx"] = np.array(delta_x).flatten()
                return _oc(Kg, T_H, R_n, delta_x)

            f.correct = wrapped
            f.update(feats)
            f.correct = orig_correct
            put(2, view_ptr=prob.view_ptr, obs_uv=prob.obs_uv, obs_slot=prob.obs_slot, idp_base=prob.idp_base,
                idp_m=prob.idp_m, idp_rho=prob.idp_rho, cam_R=cam_R, cam_t=cam_t, dx=cap.get("dx", np.zeros(15 + 6 * N)),
                status=np.int32(0 if "dx" in cap else 1), P_after=f.state.covariance.copy(),
                post_cam_R=np.stack([f.state.cameras[k].T_W_Ci.R for k in keys]),
                post_cam_t=np.stack([f.state.cameras[k].T_W_Ci.t for k in keys]),
                post_imu_R=imu.T_W_Ii.R.copy(), post_imu_t=imu.T_W_Ii.t.copy(), post_imu_v=imu.v_W_Ii.copy(),
                post_imu_bg=imu.gyroscope_bias.copy(), post_imu_ba=imu.accelerometer_bias.copy(), **pre)
        if N > max_clones:
            drop = [keys[1], keys[3]]                       # two non-adjacent clones, as prune_* would pick
            slots = [1, 3]
            f.remove_cameras({k: f.state.cameras[k] for k in drop})
            put(3, slots=np.array(slots, dtype=np.int32), P_after=f.state.covariance.copy())
    arrays["op_kind"] = np.array(kinds, dtype=np.int32)
    arrays["versions"] = np.array([np.__version__, scipy.__version__, sys.version.split()[0]])
    return arrays


def coincident_clone_problem(seed):
    """SURVEY.md section 7 hard part 4: rank(H_f) < 3.  Clones 0..3 share one position (pure rotation), so a track
    seen only from them has H_f w = 0 for the common ray w: scipy's null_space returns 2M - 2 columns there
    (`MSCKF.py:554-559`), dof = 2M - 2 at the gate (`:564`).  Tracks starting later are ordinary."""
    rng = np.random.default_rng(seed)
    N = 10
    cam_R, cam_t = synth.clone_poses(N, rng)
    cam_t[1:4] = cam_t[0]
    prob = synth.make_problem(N, 70, 4, seed=seed, poses=(cam_R, cam_t), variable_tracks=True, min_track=2)
    return prob


def gate_threshold_problem(seed, rel=1e-4):
    """SURVEY.md section 7 hard part 1: features whose gate statistic sits within `rel` of the critical value, one
    on either side.  A pixel offset on one view of two features is tuned by bisection on the REFERENCE's own
    gamma (`compute_residual_and_jacobians` + the expression of `gating_test`, `MSCKF.py:561-568`)."""
    prob = synth.make_problem(12, 60, 6, seed=seed)

    def gamma_of(j, s):
        uv = prob.obs_uv.copy()
        a = int(prob.view_ptr[j])
        uv[a] = uv[a] + s * np.array([1.0, -0.6])
        q = synth.UpdateProblem(**{**prob.__dict__, "obs_uv": uv})
        f, feats, _ = build_filter(q)
        ft = list(feats.values())[j]
        r_o, H_o = f.compute_residual_and_jacobians(ft)
        S = H_o @ f.state.covariance @ H_o.T + f.sigma_image ** 2 * np.eye(H_o.shape[0])
        return float((r_o.T @ np.linalg.inv(S) @ r_o).flatten()[0]), float(chi2.ppf(0.95, r_o.shape[0]))

    uv = prob.obs_uv.copy()
    for j, side in ((7, -1.0), (23, +1.0)):
        lo, hi = 0.0, 400.0
        g_hi, crit = gamma_of(j, hi)
        assert g_hi > crit
        target = crit * (1.0 + side * rel)
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            g, _ = gamma_of(j, mid)
            if g < target:
                lo = mid
            else:
                hi = mid
            if abs(g - target) < 0.02 * rel * crit:
                break
        a = int(prob.view_ptr[j])
        uv[a] = uv[a] + mid * np.array([1.0, -0.6])
    return synth.UpdateProblem(**{**prob.__dict__, "obs_uv": uv})


def build_filter(prob, keys=None):
    """The reference filter and its Feature dict for a flat problem (state as in run_reference, default IMU)."""
    params = MSCKFParameters()
    params.K = prob.K
    params.sigma_image = prob.sigma
    params.W_gravity = prob.gravity.copy()
    f = MSCKF(params)
    if keys is None:
        keys = [10 * (i + 1) for i in range(prob.N)]
    for i, k in enumerate(keys):
        f.state.cameras[k] = Camera(prob.K, 640, 480, Isometry3D(prob.cam_R[i].copy(), prob.cam_t[i].copy()))
    f.state.covariance = prob.P.copy()
    feats = {}
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        ft = Feature()
        ft.keypoints = [prob.obs_uv[i].copy() for i in range(a, b)]
        ft.camera_indices = [keys[int(prob.obs_slot[i])] for i in range(a, b)]
        idp = InverseDepthPoint()
        idp.base, idp.m, idp.rho = prob.idp_base[j].copy(), prob.idp_m[j].copy(), float(prob.idp_rho[j])
        ft.inverse_depth_point = idp
        feats[100 + j] = ft
    return f, feats, keys


def run_reference_prune(prob, tracks, sp, method="poorest", max_states=None):
    """`MSCKF.prune_poorest_camera_states` (`MSCKF.py:710-737`): the two clones seen by the fewest features, the
    features seen by them -> get_valid_features -> update -> remove_cameras.  Returns what it left behind.
    method="states": `MSCKF.prune_camera_states` (`:663-680`, every int(max / to_delete)-th clone) instead, with a
    `last_camera_measurement` in place so that the bookkeeping of `remove_cameras` (`:771-777`) is on record too."""
    params = MSCKFParameters()
    if max_states is not None:
        params.max_number_of_camera_states = int(max_states)
    params.K = prob.K
    params.sigma_image = prob.sigma
    params.W_gravity = prob.gravity.copy()
    params.width, params.height = sp.width, sp.height
    params.use_parallax = sp.use_parallax
    params.min_parallax = sp.min_parallax_deg
    params.min_number_of_frames_to_be_lost = sp.min_frames_lost
    params.min_number_of_frames_to_be_tracked = sp.min_frames_tracked
    f = MSCKF(params)
    keys = [10 * (i + 1) for i in range(prob.N)]
    for i, k in enumerate(keys):
        f.state.cameras[k] = Camera(prob.K, sp.width, sp.height, Isometry3D(prob.cam_R[i].copy(), prob.cam_t[i].copy()))
    f.state.covariance = prob.P.copy()
    feats = {}
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        ft = Feature()
        ft.keypoints = [prob.obs_uv[i].copy() for i in range(a, b)]
        ft.camera_indices = [keys[int(prob.obs_slot[i])] for i in range(a, b)]
        ft.lines = [Line(tracks.line_base[i].copy(), tracks.line_dir[i].copy(), float(tracks.line_conf[i])) for i in range(a, b)]
        ft.descriptors = [None] * (b - a)
        ft.scores = [0.0] * (b - a)
        ft.lost_for_n_frames = int(tracks.lost_for[j])
        ft.tracked_for_n_frames = int(tracks.tracked_for[j])
        idp = InverseDepthPoint()
        idp.base, idp.m, idp.rho = prob.idp_base[j].copy(), prob.idp_m[j].copy(), float(prob.idp_rho[j])
        ft.inverse_depth_point = idp
        feats[100 + j] = ft
    f.features = feats
    f.estimated_world_points = []
    f.currently_processed_world_points = []
    cap = {}
    orig_correct = f.correct

    def wrapped(Kg, T_H, R_n, delta_x):
        cap["dx"] = np.array(delta_x).flatten()
        return orig_correct(Kg, T_H, R_n, delta_x)

    f.correct = wrapped
    rej0 = f.number_of_residuals_discarded_for_gasting_test
    extra = {}
    if method == "states":
        from src.msckf.FeatureExtractor import CameraMeasurement
        ids = np.array(sorted(feats.keys()), dtype=np.int64)
        f.last_camera_measurement = CameraMeasurement(keypoints=[], descriptors=np.arange(len(ids) * 4, dtype=np.float64).reshape(len(ids), 4),
                                                      scores=[], features_indices=ids.copy())
        f.prune_camera_states()
        extra = dict(prune_lcm_indices_left=np.asarray(f.last_camera_measurement.features_indices, dtype=np.int64),
                     prune_lcm_descriptors_left=np.asarray(f.last_camera_measurement.descriptors, dtype=np.float64),
                     prune_max_states=np.int32(f.max_number_of_camera_states),
                     prune_states_to_delete=np.int32(f.camera_states_to_delete))
    else:
        f.prune_poorest_camera_states()
    left = list(f.state.cameras.keys())
    removed = [i for i, k in enumerate(keys) if k not in left]
    views_left = np.array([len(feats[100 + j].camera_indices) if (100 + j) in f.features else 0 for j in range(prob.F)], dtype=np.int32)
    return dict(extra, prune_removed_slots=np.array(removed, dtype=np.int32), prune_P_after=f.state.covariance.copy(),
                prune_dx=cap.get("dx", np.zeros(prob.d)), prune_status=np.int32(0 if "dx" in cap else 1),
                prune_n_rejected=np.int32(f.number_of_residuals_discarded_for_gasting_test - rej0),
                prune_post_cam_R=np.stack([f.state.cameras[k].T_W_Ci.R for k in left]),
                prune_post_cam_t=np.stack([f.state.cameras[k].T_W_Ci.t for k in left]),
                prune_views_left=views_left, prune_features_left=np.int32(len(f.features)))


def run_reference_associate(seed):
    """f4: the reference's own `add_camera_measurements` (`MSCKF.py:268-448`) on prepared matches.  The tracks live
    in clones 0..N-2, the newest clone N-1 is `state.cameras[state.imu.id]`; the descriptor matcher (XFeat, absent
    here) is replaced by a function that returns the prepared (feature, keypoint) pairs, everything else -- the
    per-view epipolar / homography tests, the counters, the appended views -- is the reference's code."""
    from src.msckf.FeatureExtractor import CameraMeasurement, ExtractedFeature
    rng = np.random.default_rng(seed)
    N = 9
    cam_R, cam_t = synth.clone_poses(N, rng)
    cam_t[N - 1] = cam_t[3] + 1e-3 * rng.standard_normal(3)            # the newest clone sits 1 mm from clone 3: homography test there
    prob = synth.make_problem(N - 1, 160, 6, seed=seed, poses=(cam_R[:N - 1], cam_t[:N - 1]), variable_tracks=True, min_track=1)
    params = MSCKFParameters()
    params.K = prob.K
    # thresholds in the units of the reference's scores (normalised epipolar form, pixels for the homography)
    params.epipolar_rejection_threshold = 2e-4
    params.homography_rejection_threshold = 1.5
    f = MSCKF(params)
    keys = [10 * (i + 1) for i in range(N)]
    for i, k in enumerate(keys):
        f.state.cameras[k] = Camera(prob.K, 640, 480, Isometry3D(cam_R[i].copy(), cam_t[i].copy()))
    f.state.imu.id = keys[-1]
    feats = {}
    matched_uv = np.full((prob.F, 2), np.nan)
    Kf = np.asarray(prob.K, dtype=np.float64)
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        ft = Feature()
        ft.keypoints = [prob.obs_uv[i].copy() for i in range(a, b)]
        ft.camera_indices = [keys[int(prob.obs_slot[i])] for i in range(a, b)]
        ft.descriptors = [rng.standard_normal(4) for _ in range(a, b)]
        ft.scores = [1.0] * (b - a)
        ft.lines = [Line(cam_t[int(prob.obs_slot[i])], np.array([0.0, 0.0, 1.0]), 1.0) for i in range(a, b)]
        ft.tracked_for_n_frames = b - a
        feats[100 + j] = ft
        if rng.uniform() < 0.85:                                       # 15 % of the features find no match in the new image
            pw = prob.idp_base[j] + prob.idp_m[j] / prob.idp_rho[j]    # a 3-D point consistent with the track (up to its noise)
            q = cam_R[N - 1].T @ (pw - cam_t[N - 1])
            px = Kf @ q
            px = px[:2] / px[2]
            px = px + 0.3 * rng.standard_normal(2)
            if rng.uniform() < 0.3:
                px = px + rng.uniform(-40, 40, 2)                      # wrong matches
            matched_uv[j] = px
    f.features = feats
    ids = [100 + j for j in range(prob.F) if not np.isnan(matched_uv[j, 0])]
    lost_ids = [100 + j for j in range(prob.F) if np.isnan(matched_uv[j, 0])]

    def match(last, cur, min_cos):
        m = CameraMeasurement(keypoints=[matched_uv[i - 100].copy() for i in ids], descriptors=[rng.standard_normal(4) for _ in ids],
                              scores=[1.0] * len(ids), features_indices=list(ids))
        nm = CameraMeasurement(keypoints=[], descriptors=[], scores=[], features_indices=list(lost_ids))
        return m, nm, None

    f.feature_extractor = types.SimpleNamespace(match=match)
    f.last_camera_measurement = CameraMeasurement()
    f.current_image = np.zeros((4, 4, 3), dtype=np.uint8)
    n_views0 = np.array([len(feats[100 + j].keypoints) for j in range(prob.F)])
    kp = [matched_uv[i - 100].copy() for i in ids]
    f.add_camera_measurements(np.zeros((4, 4, 3), dtype=np.uint8),
                              ExtractedFeature(keypoints=kp, descriptors=[rng.standard_normal(4) for _ in ids], scores=[1.0] * len(ids)))
    kept = np.array([len(feats[100 + j].keypoints) - n_views0[j] for j in range(prob.F)], dtype=np.uint8)
    out = dict(assoc_matched_uv=matched_uv, assoc_R_cur=cam_R[N - 1], assoc_t_cur=cam_t[N - 1],
               assoc_thr=np.array([params.epipolar_rejection_threshold, params.homography_rejection_threshold]),
               assoc_kept=kept, assoc_n_epipolar=np.int32(f.number_of_features_discarded_for_epipolar_test),
               assoc_n_homography=np.int32(f.number_of_features_discarder_for_homography_test),
               assoc_lost_for=np.array([feats[100 + j].lost_for_n_frames for j in range(prob.F)], dtype=np.int32))
    return prob, out


def save(name, prob, out):
    arrays = dict(
        P=prob.P, cam_R=prob.cam_R, cam_t=prob.cam_t, cam_R0=prob.cam_R0, cam_t0=prob.cam_t0,
        gravity=prob.gravity, K=prob.K, sigma=np.float64(prob.sigma), view_ptr=prob.view_ptr,
        obs_uv=prob.obs_uv, obs_slot=prob.obs_slot, idp_base=prob.idp_base, idp_m=prob.idp_m,
        idp_rho=prob.idp_rho,
        versions=np.array([np.__version__, scipy.__version__, sys.version.split()[0]]))
    arrays.update(out)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    acc = int(out["accepted"].sum())
    print(f"{name:28s} N={prob.N:3d} F={prob.F:5d} accepted={acc:5d} status={int(out['status'])} "
          f"|dx|={np.linalg.norm(out['dx']):.3e} margin={float(out['min_gate_margin']):.2e} "
          f"Rn_dev={float(out['Rn_dev']):.1e} size={os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--headline", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    cases = {}
    # config 1 and 2, recipe A (random SPD P) and B (P and poses from the reference's own propagation)
    cases["cfg1_A"] = lambda: (synth.make_problem(10, 50, 5, seed=0), None)
    cases["cfg2_A"] = lambda: (synth.make_problem(20, 500, 8, seed=1), None)

    def recipe_b(N, F, M, seed):
        P, cam_R, cam_t, keys = realistic_state(N, seed)
        prob = synth.make_problem(N, F, M, seed=seed, P=P, poses=(cam_R, cam_t))
        return prob, keys
    cases["cfg1_B"] = lambda: recipe_b(10, 50, 5, 2)
    cases["cfg2_B"] = lambda: recipe_b(20, 500, 8, 3)
    # edge cases
    cases["edge_variable_tracks"] = lambda: (synth.make_problem(12, 80, 12, seed=4, variable_tracks=True, min_track=2), None)
    cases["edge_no_qr_branch"] = lambda: (synth.make_problem(10, 6, 5, seed=5), None)            # m = 42 <= d = 75
    cases["edge_all_rejected"] = lambda: (synth.make_problem(8, 20, 5, seed=6, sigma=0.01, pixel_noise=80.0), None)   # every gate fails -> no-op
    cases["edge_some_rejected"] = lambda: (synth.make_problem(12, 120, 6, seed=7, outlier_fraction=0.25, outlier_px=500.0), None)
    cases["edge_null_pose"] = lambda: (synth.make_problem(10, 60, 6, seed=8, distinct_null=True), None)
    cases["edge_zero_gravity"] = lambda: (synth.make_problem(10, 60, 6, seed=9, gravity=np.zeros(3)), None)
    cases["edge_int_K"] = lambda: (synth.make_problem(10, 40, 5, seed=10, K=np.array([[180, 0, 320], [0, 180, 240], [0, 0, 1]])), None)
    cases["edge_sigma_01"] = lambda: (synth.make_problem(15, 150, 7, seed=11, sigma=0.1, outlier_fraction=0.1, outlier_px=300.0), None)
    cases["edge_single_feature"] = lambda: (synth.make_problem(6, 1, 4, seed=12), None)
    cases["edge_full_window_tracks"] = lambda: (synth.make_problem(8, 40, 8, seed=13), None)      # every track spans all clones
    cases["edge_rank2_Hf"] = lambda: (coincident_clone_problem(14), None)                         # tracks with rank(H_f) = 2
    cases["edge_gate_threshold"] = lambda: (gate_threshold_problem(15), None)                     # gamma within 1e-4 of crit, both sides
    cases["edge_gate_threshold_tight"] = lambda: (gate_threshold_problem(16, rel=1e-7), None)     # ... within 1e-7: the reference decides the mask
    # Round 5: long tracks, pinned to the reference itself (the reference's window is 30 clones, MSCKF.py:45; tracks grow a view
    # per frame until lost, :404-412; both pruning callers hand update every feature of the removed clones, :669-678, :726-735).
    # Until now the longest fixture track had 12 views: k_feature<32> / <64>, the split of long tracks and the dense
    # remainder were compared with the oracle only.
    cases["edge_long_tracks"] = lambda: (synth.make_problem(31, 64, 31, seed=24), None)           # every track spans the whole 31-clone window
    cases["edge_mixed_spans"] = lambda: (synth.make_problem(30, 300, 30, seed=41, variable_tracks=True, min_track=2,
                                                            outlier_fraction=0.10, outlier_px=400.0), None)
    cases["edge_few_long_among_short"] = lambda: (synth.few_long_tracks_problem(30, 400, 10, 10, seed=42), None)

    def long_recipe_b():
        P, cam_R, cam_t, keys = realistic_state(30, 43)
        return synth.make_problem(30, 160, 30, seed=43, P=P, poses=(cam_R, cam_t), variable_tracks=True, min_track=2), keys
    cases["edge_long_tracks_B"] = long_recipe_b

    def gauge_prior():
        # metre-level COMMON-MODE variance of the clone (and IMU) positions: the stacked Jacobian's null space (global
        # translation and yaw, rank 6N - 4) is where this prior is largest -- any shortcut that adds information there
        # (a shifted normal-equation form) shows up in P+ at once
        prob = synth.make_problem(30, 150, 30, seed=44, variable_tracks=True, min_track=2)
        d = prob.d
        U = np.zeros((d, 3))
        U[12:15] = np.eye(3)
        for i in range(prob.N):
            U[18 + 6 * i:21 + 6 * i] = np.eye(3)
        P = prob.P + 100.0 * (U @ U.T)                                                            # sigma = 10 m, fully correlated
        return synth.UpdateProblem(**{**prob.__dict__, "P": P}), None
    cases["edge_gauge_prior"] = gauge_prior
    cases["edge_few_rows_long_tracks"] = lambda: (synth.make_problem(20, 5, 18, seed=45, variable_tracks=True, min_track=14), None)   # m << 6N
    if args.headline:
        cases["cfg3_A"] = lambda: (synth.make_problem(30, 2000, 10, seed=0), None)

    for name, mk in cases.items():
        if args.only and name != args.only:
            continue
        prob, keys = mk()
        out = run_reference(prob, keys)
        save(name, prob, out)

    # f1: get_valid_features (+ the chained update), SURVEY.md §8(f1)
    SP = synth.SelectParams
    sel_cases = {}
    sel_cases["sel_default"] = lambda: (synth.make_problem(12, 150, 8, seed=20), dict(seed=20), SP(), None)
    sel_cases["sel_parallax5"] = lambda: (synth.make_problem(20, 400, 10, seed=21, outlier_fraction=0.1, outlier_px=400.0), dict(seed=21, lost_fraction=0.3),
                                          SP(min_parallax_deg=5.0), None)
    sel_cases["sel_no_parallax"] = lambda: (synth.make_problem(10, 80, 6, seed=22), dict(seed=22), SP(use_parallax=False), None)
    sel_cases["sel_behind_camera"] = lambda: (synth.make_problem(10, 80, 6, seed=23), dict(seed=23, flip_fraction=0.3),
                                              SP(min_parallax_deg=2.0, min_frames_tracked=3), None)
    sel_cases["sel_variable_tracks"] = lambda: (synth.make_problem(12, 120, 12, seed=24, variable_tracks=True, min_track=1),
                                                dict(seed=24, lost_fraction=0.8), SP(min_frames_tracked=1), None)
    sel_cases["sel_none_valid"] = lambda: (synth.make_problem(8, 30, 5, seed=25), dict(seed=25, lost_fraction=0.0), SP(), None)
    sel_cases["sel_small_image"] = lambda: (synth.make_problem(10, 100, 6, seed=26), dict(seed=26, lost_fraction=0.9),
                                            SP(min_frames_tracked=2, width=330, height=250), None)

    def sel_recipe_b():
        P, cam_R, cam_t, keys = realistic_state(15, 27)
        prob = synth.make_problem(15, 200, 8, seed=27, P=P, poses=(cam_R, cam_t))
        return prob, dict(seed=27), SP(min_parallax_deg=4.0), keys
    sel_cases["sel_recipe_B"] = sel_recipe_b
    for name, kw in {"seq_short": dict(seed=1, frames=5, imu_per_frame=6, F=30, M=4, max_clones=4),
                     "seq_long": dict(seed=2, frames=12, imu_per_frame=8, F=60, M=6, max_clones=8)}.items():
        if args.only and name != args.only:
            continue
        arrays = run_reference_sequence(**kw)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        k = arrays["op_kind"]
        print(f"{name:28s} ops={len(k)} imu={int((k == 0).sum())} augment={int((k == 1).sum())} "
              f"update={int((k == 2).sum())} remove={int((k == 3).sum())} size={os.path.getsize(path) / 1024:.0f} KiB", flush=True)

    for name, mk in sel_cases.items():
        if args.only and name != args.only:
            continue
        prob, tk, sp, keys = mk()
        tracks = synth.make_tracks(prob, **tk)
        save_select(name, prob, tracks, sp, run_reference_select(prob, tracks, sp, keys))

    # prune_poorest_camera_states (MSCKF.py:710-737): select -> update -> remove_cameras back to back.  Stored like a
    # sel_* case (the selection outputs are those of the features the prune hands to get_valid_features).
    if not args.only or args.only == "sel_prune_poorest":
        prob = synth.make_problem(10, 90, 6, seed=30, variable_tracks=True, min_track=2)
        sp = SP(min_parallax_deg=3.0, min_frames_tracked=2)
        tracks = synth.make_tracks(prob, seed=30, lost_fraction=0.7)
        out = run_reference_prune(prob, tracks, sp)
        sel = run_reference_select(prob, tracks, sp, None)        # (for the loader: the whole-dict selection + update)
        sel.update(out)
        save_select("sel_prune_poorest", prob, tracks, sp, sel)
        print("    prune removed slots", out["prune_removed_slots"], "status", int(out["prune_status"]),
              "features left", int(out["prune_features_left"]))

    # prune_camera_states (MSCKF.py:663-680): every int(max / to_delete)-th clone of the window; tracks short enough
    # that some features lose ALL their views with the removed clones (remove_cameras :770-777 then touches
    # last_camera_measurement)
    if not args.only or args.only == "sel_prune_states":
        prob = synth.make_problem(12, 110, 5, seed=31, variable_tracks=True, min_track=1)
        sp = SP(min_parallax_deg=3.0, min_frames_tracked=2)
        tracks = synth.make_tracks(prob, seed=31, lost_fraction=0.7)
        out = run_reference_prune(prob, tracks, sp, method="states", max_states=12)
        sel = run_reference_select(prob, tracks, sp, None)
        sel.update(out)
        save_select("sel_prune_states", prob, tracks, sp, sel)
        print("    prune_camera_states removed slots", out["prune_removed_slots"], "status", int(out["prune_status"]),
              "features left", int(out["prune_features_left"]), "lcm entries left", len(out["prune_lcm_indices_left"]))

    # f4: the per-view consistency tests of add_camera_measurements (MSCKF.py:332-412)
    if not args.only or args.only == "assoc_tests":
        prob, out = run_reference_associate(40)
        arrays = dict(P=prob.P, cam_R=prob.cam_R, cam_t=prob.cam_t, cam_R0=prob.cam_R0, cam_t0=prob.cam_t0, gravity=prob.gravity,
                      K=prob.K, sigma=np.float64(prob.sigma), view_ptr=prob.view_ptr, obs_uv=prob.obs_uv, obs_slot=prob.obs_slot,
                      idp_base=prob.idp_base, idp_m=prob.idp_m, idp_rho=prob.idp_rho,
                      versions=np.array([np.__version__, scipy.__version__, sys.version.split()[0]]))
        arrays.update(out)
        path = os.path.join(HERE, "assoc_tests.npz")
        np.savez_compressed(path, **arrays)
        k = out["assoc_kept"]
        print(f"{'assoc_tests':28s} N={prob.N:3d} F={prob.F:5d} matched={int((~np.isnan(out['assoc_matched_uv'][:, 0])).sum())} kept={int(k.sum())} "
              f"epipolar={int(out['assoc_n_epipolar'])} homography={int(out['assoc_n_homography'])} size={os.path.getsize(path) / 1024:.0f} KiB")

    if not args.only:
        table = np.array([0.0] + [chi2.ppf(0.95, k) for k in range(1, 513)])
        np.save(os.path.join(HERE, "chi2_ppf_095.npy"), table)
        print("chi2 table written:", table[:4], "...")


if __name__ == "__main__":
    main()
