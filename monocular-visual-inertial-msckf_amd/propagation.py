"""Host-side pieces of the steps either side of the update.

The covariance arithmetic of `MSCKF.process_imu` (reference `src/msckf/MSCKF.py:236-244`),
`state_augmentation` (`:262-265`) and `remove_cameras` (`:754-757`) runs on the device
(`msckf_propagate`, `msckf_augment`, `msckf_remove_clones`); what stays here is the 15x15 /
6x15 set-up those calls take as arguments, built from the IMU state exactly as the reference
does.  NumPy only."""
from __future__ import annotations

import numpy as np


def _hat(w):
    """3x3 cross-product matrix (reference `src/utils/geometry.py:222-235`)."""
    x, y, z = (float(w[0]), float(w[1]), float(w[2]))
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def imu_transition(R, t, v, R0, t0, v0, gyro, acc, dt, gravity, noise, planet_rate=None):
    """Discrete transition `Phi` (15x15) with the observability constraint, and the discrete
    noise `Q`, of one `process_imu` step (reference `MSCKF.py:179-237`).

    `R, t, v`: IMU orientation / position / velocity after `imu.integrate` (`:168`);
    `R0, t0, v0`: the null state (`:221-230`); `gyro, acc`: bias-corrected sample (`:166-167`);
    `noise`: the 12x12 continuous noise covariance (`:98-103`).
    Error-state order: [dtheta, db_g, dv, db_a, dp] (`:171`)."""
    R, R0 = np.asarray(R, dtype=np.float64), np.asarray(R0, dtype=np.float64)
    g = np.asarray(gravity, dtype=np.float64)
    wp = np.zeros(3) if planet_rate is None else np.asarray(planet_rate, dtype=np.float64)
    I3, Wp = np.eye(3), _hat(wp)
    Z = np.zeros((3, 3))
    F = np.block([
        [-_hat(gyro), -I3, Z, Z, Z],                                  # :182-183
        [Z, Z, Z, Z, Z],
        [-R @ _hat(acc), Z, -2.0 * Wp, -R, -Wp @ -Wp],                # :186-189
        [Z, Z, Z, Z, Z],
        [Z, Z, I3, Z, Z]])                                            # :192
    G = np.block([
        [-I3, Z, Z, Z],                                               # :203
        [Z, I3, Z, Z],                                                # :206
        [Z, Z, -R, Z],                                                # :209
        [Z, Z, Z, I3],                                                # :212
        [Z, Z, Z, Z]])
    A = F * dt
    A2 = A @ A
    Phi = np.eye(15) + A + 0.5 * A2 + (1.0 / 6.0) * (A2 @ A)          # :215-218
    Phi[0:3, 0:3] = R @ R0.T                                          # :221
    u = R0 @ g                                                        # :223
    s = u / (u @ u)                                                   # :224
    w1 = _hat(np.asarray(v0) - np.asarray(v)) @ g                     # :229
    w2 = _hat(dt * np.asarray(v0) + np.asarray(t0) - np.asarray(t)) @ g   # :230
    for rows, w in ((slice(6, 9), w1), (slice(12, 15), w2)):          # :226-233
        blk = Phi[rows, 0:3].copy()
        Phi[rows, 0:3] = blk - np.outer(blk @ u - w, s)
    Q = Phi @ G @ np.asarray(noise, dtype=np.float64) @ G.T @ Phi.T * dt   # :237
    return Phi, Q


def augmentation(imu_R, imu_t, T_W_I, T_W_C):
    """Pose of the new clone and the non-zero 6x15 block of the augmentation Jacobian
    (reference `MSCKF.py:252-261`).  `T_W_I`, `T_W_C`: (R, t) of the static IMU and camera
    frames (`IMU.py:26`, `MSCKF.py:89`)."""
    def hom(R, t):
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
        return T
    T_I_C = np.linalg.inv(hom(*T_W_I)) @ hom(*T_W_C)                  # :252 (Isometry3D.inv / __mul__)
    T_W_Ci = hom(imu_R, imu_t) @ T_I_C                                # :253
    J = np.zeros((6, 15))
    J[0:3, 0:3] = T_I_C[:3, :3].T                                     # :259
    J[3:6, 0:3] = _hat(np.asarray(imu_R, dtype=np.float64) @ T_I_C[:3, 3])   # :260
    J[3:6, 12:15] = np.eye(3)                                         # :261
    return J, T_W_Ci[:3, :3].copy(), T_W_Ci[:3, 3].copy()
