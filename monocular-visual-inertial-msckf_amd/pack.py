"""Reference-shaped objects -> the flat arrays of the C-ABI.

Reads exactly what `MSCKF.update` reads (reference `src/msckf/MSCKF.py:497-552`):
per feature `keypoints`, `camera_indices`, `inverse_depth_point.{base,m,rho}`;
per camera `T_W_Ci.{R,t}` and `T_W_Ci_null.{R,t}`; `state.covariance`,
`state.imu.W_gravity`, `K`, `sigma_image`.  A clone's column slot is its
position in the ordered `cameras` mapping at call time (`MSCKF.py:539`)."""
from __future__ import annotations

import numpy as np

from .synth import SelectParams, TrackTable, UpdateProblem


def problem_from_reference(filt, features) -> UpdateProblem:
    cams = filt.state.cameras
    keys = list(cams.keys())
    slot_of = {k: i for i, k in enumerate(keys)}
    N = len(keys)
    cam_R = np.empty((N, 3, 3))
    cam_t = np.empty((N, 3))
    cam_R0 = np.empty((N, 3, 3))
    cam_t0 = np.empty((N, 3))
    for i, k in enumerate(keys):
        c = cams[k]
        cam_R[i] = c.T_W_Ci.R
        cam_t[i] = c.T_W_Ci.t
        cam_R0[i] = c.T_W_Ci_null.R
        cam_t0[i] = c.T_W_Ci_null.t
    view_ptr = [0]
    uv, slots, base, m, rho = [], [], [], [], []
    for ft in features.values():                       # dict order, MSCKF.py:573
        for kp, ci in zip(ft.keypoints, ft.camera_indices):
            uv.append(np.asarray(kp, dtype=np.float64)[:2])
            slots.append(slot_of[ci])
        view_ptr.append(len(slots))
        idp = ft.inverse_depth_point
        base.append(np.asarray(idp.base, dtype=np.float64))
        m.append(np.asarray(idp.m, dtype=np.float64))
        rho.append(float(idp.rho))
    F = len(rho)
    return UpdateProblem(
        P=np.array(filt.state.covariance, dtype=np.float64), cam_R=cam_R, cam_t=cam_t, cam_R0=cam_R0, cam_t0=cam_t0,
        gravity=np.asarray(filt.state.imu.W_gravity, dtype=np.float64), K=np.asarray(filt.K),
        sigma=float(filt.sigma_image), view_ptr=np.asarray(view_ptr, dtype=np.int32),
        obs_uv=np.asarray(uv, dtype=np.float64).reshape(-1, 2), obs_slot=np.asarray(slots, dtype=np.int32),
        idp_base=np.asarray(base, dtype=np.float64).reshape(F, 3), idp_m=np.asarray(m, dtype=np.float64).reshape(F, 3),
        idp_rho=np.asarray(rho, dtype=np.float64), meta={"keys": keys})


def tracks_from_reference(features) -> TrackTable:
    """What `MSCKF.get_valid_features` reads besides the views (reference
    `MSCKF.py:461-480`): `Feature.lines[i].{base,direction,confidence}` and the frame
    counters `lost_for_n_frames`, `tracked_for_n_frames`."""
    base, direction, conf, lost, tracked = [], [], [], [], []
    for ft in features.values():
        if len(ft.lines) != len(ft.keypoints):
            raise ValueError("a feature needs one line per view (reference MSCKF.py:410, :765-769)")
        for ln in ft.lines:
            base.append(np.asarray(ln.base, dtype=np.float64).reshape(3))
            direction.append(np.asarray(ln.direction, dtype=np.float64).reshape(3))
            conf.append(float(ln.confidence))
        lost.append(int(ft.lost_for_n_frames))
        tracked.append(int(ft.tracked_for_n_frames))
    return TrackTable(line_base=np.asarray(base, dtype=np.float64).reshape(-1, 3),
                      line_dir=np.asarray(direction, dtype=np.float64).reshape(-1, 3),
                      line_conf=np.asarray(conf, dtype=np.float64), lost_for=np.asarray(lost, dtype=np.int32),
                      tracked_for=np.asarray(tracked, dtype=np.int32))


def select_params_from_reference(filt) -> SelectParams:
    """The filter attributes `get_valid_features` reads (`MSCKF.py:114-120`; image size from the
    cameras, `Camera.py:24-25`)."""
    cam = next(iter(filt.state.cameras.values()))
    return SelectParams(min_frames_lost=int(filt.min_number_of_frames_to_be_lost),
                        min_frames_tracked=int(filt.min_number_of_frames_to_be_tracked),
                        use_parallax=bool(filt.use_parallax), min_parallax_deg=float(filt.min_parallax),
                        width=int(cam.width), height=int(cam.height))
