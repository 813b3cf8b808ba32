"""State injection half of `MSCKF.correct` (reference `src/msckf/MSCKF.py:616-661`).
N+1 3x3 exp-maps with an SVD clean-up and additive corrections; host-side NumPy
(component #2).  Mutates the state in place like the
reference does, including the in-place `+=` on the translation arrays."""
from __future__ import annotations

import numpy as np


def _skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def corrected_rotation(R_est, dtheta):
    """R <- R Exp(dtheta)^T, then project on SO(3) by SVD (`MSCKF.py:625-635`)."""
    n = np.linalg.norm(dtheta)
    S = _skew(dtheta)
    if np.isclose(n, 0):
        E = np.eye(3)
    else:
        E = np.eye(3) + (np.sin(n) / n) * S + ((1 - np.cos(n)) / n ** 2) * (S @ S)
    U, _, Vt = np.linalg.svd(R_est @ E.T)
    return U @ Vt


def inject_state(state, dx):
    dx = np.asarray(dx, dtype=np.float64).reshape(-1)
    imu = state.imu
    imu.T_W_Ii.R = corrected_rotation(imu.T_W_Ii.R, dx[0:3])          # :625-635
    imu.T_W_Ii.t += dx[12:15]                                         # :637
    imu.v_W_Ii += dx[6:9]                                             # :638
    imu.gyroscope_bias += dx[3:6]                                     # :639
    imu.accelerometer_bias += dx[9:12]                                # :640
    for i, (_, cam) in enumerate(state.cameras.items()):              # :643
        dc = dx[15 + 6 * i: 21 + 6 * i]
        cam.T_W_Ci.R = corrected_rotation(cam.T_W_Ci.R, dc[:3])       # :649-660
        cam.T_W_Ci.t += dc[3:6]                                       # :661
