"""CPU oracle for the MSCKF measurement-update path -- TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy/SciPy restatement of the reference algorithm
(`/root/reference/src/msckf/MSCKF.py:497-661`, `src/msckf/Camera.py:38-67`,
`src/utils/geometry.py:222-235`) and of the step before it, `get_valid_features`
(`MSCKF.py:458-495`, `geometry.py:237-303`, `Camera.py:13-52`).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product path
(`monocular-visual-inertial-msckf_amd/`) never does and fails loudly when the
HIP library is missing.

Parity status: PINNED.  The reference holds no tests or golden vectors of its
own, so this oracle is pinned against outputs of the reference
itself: `tests/golden/gen_golden.py` imports `/root/reference` (with stub
modules for cv2/rerun/IPython/XFeat), drives `MSCKF.update` on seeded synthetic
problems and stores inputs and outputs as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this file against every one of them.

Every function takes the flat arrays of `UpdateProblem` (see
`monocular-visual-inertial-msckf_amd/synth.py`), all float64.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import null_space
from scipy.stats import chi2


def skew(w):
    """reference `src/utils/geometry.py:222-235`."""
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def view_terms(R_WC, t_WC, R0_WC, t0_WC, rho, base, m, uv, K, g):
    """One view of one feature: residual (2,), OC-projected clone block A (2,6),
    feature block H_f (2,3).  reference `MSCKF.py:505-544` + `Camera.py:54-67`."""
    R_CW = R_WC.T                                           # MSCKF.py:507
    R_CW0 = R0_WC.T                                         # :509
    Ci_f = R_CW @ (rho * (base - t_WC) + m)                 # :516 (rho-scaled point)
    W_f = R_WC @ Ci_f + t_WC                                # :517, Camera.py:38-44
    z = np.linalg.inv(K) @ np.append(uv, 1.0)               # :519
    z = z[:2] / z[2]                                        # :520
    z_hat = np.array([Ci_f[0] / Ci_f[2], Ci_f[1] / Ci_f[2]])  # :522
    r = z - z_hat                                           # :524
    x, y, zz = Ci_f
    J = np.array([[1.0 / zz, 0.0, -x / zz ** 2], [0.0, 1.0 / zz, -y / zz ** 2]])  # Camera.py:57-58
    H_x = np.zeros((2, 6))
    H_x[:, :3] = J @ skew(Ci_f)                             # Camera.py:65
    H_x[:, 3:] = -J @ R_CW                                  # Camera.py:66
    u = np.zeros(6)
    u[:3] = R_CW0 @ g                                       # MSCKF.py:529
    u[3:] = skew(W_f - t0_WC) @ g                           # :530
    A = H_x.copy()
    den = u @ u
    if den > 1e-6:                                          # :534
        A = A - (A @ u)[:, None] * u / den
    H_f = -H_x[:, 3:]                                       # :536 (from the un-projected H_x)
    return r, A, H_f


def feature_blocks(prob, j):
    """Stacked r (2M,), dense H_x (2M, d), H_f (2M, 3) of feature j.
    reference `MSCKF.py:497-548`."""
    a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
    d = prob.P.shape[0]
    M = b - a
    r = np.zeros(2 * M)
    H_x = np.zeros((2 * M, d))
    H_f = np.zeros((2 * M, 3))
    for i in range(M):
        s = int(prob.obs_slot[a + i])
        ri, A, Hfi = view_terms(prob.cam_R[s], prob.cam_t[s], prob.cam_R0[s], prob.cam_t0[s],
                                prob.idp_rho[j], prob.idp_base[j], prob.idp_m[j], prob.obs_uv[a + i],
                                prob.K, prob.gravity)
        r[2 * i:2 * i + 2] = ri
        H_x[2 * i:2 * i + 2, 15 + 6 * s:21 + 6 * s] = A    # MSCKF.py:538-540
        H_f[2 * i:2 * i + 2] = Hfi
    return r, H_x, H_f


def project_on_nullspace(H_f, r, H_x):
    """reference `MSCKF.py:554-559` (scipy.linalg.null_space of H_f^T)."""
    A = null_space(H_f.T)
    return A.T @ r, A.T @ H_x


def gate(r_o, H_o, P, sigma):
    """reference `MSCKF.py:561-568`; returns (passed, gamma, critical value)."""
    S_inv = np.linalg.inv(H_o @ P @ H_o.T + sigma ** 2 * np.eye(H_o.shape[0]))
    gamma = float(r_o @ S_inv @ r_o)
    crit = float(chi2.ppf(0.95, r_o.shape[0]))
    return gamma <= crit, gamma, crit


def update(prob, dense_noise: bool = False):
    """The whole `MSCKF.update` + covariance half of `MSCKF.correct`
    (reference `MSCKF.py:570-614`).

    dense_noise=True allocates R_o = sigma^2 * eye(m) and forms Q^T R_o Q exactly
    as the reference does (`:589, :598`); False uses R_n = sigma^2 I analytically
    (identical to 4e-17) so large m stay runnable.

    Returns dict(status, dx, P_new, accepted, gamma, crit, T_H, r_n, n_rejected).
    status 0 = updated, 1 = no-op (nothing accepted; reference early returns
    `:584-585, :591-592`)."""
    P = prob.P
    d = P.shape[0]
    F = prob.F
    sigma = prob.sigma
    accepted = np.zeros(F, dtype=np.uint8)
    gammas = np.zeros(F)
    crits = np.zeros(F)
    H_list, r_list = [], []
    for j in range(F):                                       # MSCKF.py:573
        r, H_x, H_f = feature_blocks(prob, j)
        r_o, H_o = project_on_nullspace(H_f, r, H_x)
        ok, gammas[j], crits[j] = gate(r_o, H_o, P, sigma)
        if not ok:
            continue                                         # :577-579
        accepted[j] = 1
        H_list.append(H_o)
        r_list.append(r_o)
    out = dict(status=1, dx=np.zeros(d), P_new=P.copy(), accepted=accepted, gamma=gammas, crit=crits,
               T_H=None, r_n=None, n_rejected=int(F - accepted.sum()))
    if not H_list:                                           # :584-585
        return out
    H_X = np.vstack(H_list)
    r_o = np.concatenate(r_list)
    m = H_X.shape[0]
    if m == 0:                                               # :591-592
        return out
    if m > d:                                                # :594-598
        Q, R = np.linalg.qr(H_X, mode="reduced")
        T_H = R
        r_n = Q.T @ r_o
        if dense_noise:
            R_n = Q.T @ (sigma ** 2 * np.eye(m)) @ Q
        else:
            R_n = sigma ** 2 * np.eye(d)
    else:                                                    # :599-602
        T_H, r_n = H_X, r_o
        R_n = sigma ** 2 * np.eye(m)
    S = T_H @ P @ T_H.T + R_n                                # :605
    Kg = P @ T_H.T @ np.linalg.inv(S)                        # :606
    dx = Kg @ r_n                                            # :607
    I = np.eye(d)
    Pn = (I - Kg @ T_H) @ P @ (I - Kg @ T_H).T + Kg @ R_n @ Kg.T  # :613
    Pn = (Pn + Pn.T) / 2                                     # :614
    out.update(status=0, dx=dx, P_new=Pn, T_H=T_H, r_n=r_n, H_X=H_X, r_o=r_o)
    return out


# ---- f1: MSCKF.get_valid_features (the step before update) ------------------------------------
FLAG_VALID, FLAG_LOST, FLAG_REFRESHED = 1, 2, 4


def angle_between_directions(d1, d2):
    """reference `src/utils/geometry.py:237-256`."""
    d1 = d1 / np.linalg.norm(d1)
    d2 = d2 / np.linalg.norm(d2)
    return np.arccos(np.clip(np.dot(d1, d2), -1.0, 1.0))


def intersection_of_lines(base, direction, conf):
    """Weighted least-squares meeting point of lines, reference `geometry.py:274-303`."""
    X = np.zeros((3, 3))
    y = np.zeros(3)
    for b, dvec, c in zip(base, direction, conf):
        dn = dvec / np.linalg.norm(dvec)                    # :291
        Pm = np.eye(3) - np.outer(dn, dn)                   # :294
        X += c * Pm                                         # :296
        y += c * Pm @ b                                     # :297
    return np.linalg.pinv(X) @ y, X                         # :299


def select_features(prob, tracks, params):
    """`MSCKF.get_valid_features` (`MSCKF.py:458-495`) on the flat arrays.  Returns flags
    (bit 0 valid, bit 1 lost, bit 2 inverse-depth point refreshed), the inverse-depth points
    after the call, the triangulated world points (NaN where none was computed) and cond(X)."""
    F = prob.F
    K = np.asarray(prob.K, dtype=np.float64)
    Kinv = np.linalg.inv(K)
    min_lost = max(int(params.min_frames_lost), 1)          # the constructor's clamps, MSCKF.py:119
    min_tracked = max(int(params.min_frames_tracked), 2)    # :120
    flags = np.zeros(F, dtype=np.uint8)
    idp_m, idp_rho = prob.idp_m.copy(), prob.idp_rho.copy()
    world = np.full((F, 3), np.nan)
    cond = np.ones(F)
    for j in range(F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        lost = tracks.lost_for[j] >= min_lost                       # :463-465
        if lost and tracks.tracked_for[j] < min_tracked:            # :467-469
            flags[j] = FLAG_LOST
            continue
        enough = False
        if params.use_parallax and b - a > 1:                                     # :472
            par = np.rad2deg(angle_between_directions(tracks.line_dir[a], tracks.line_dir[b - 1]))
            enough = par > params.min_parallax_deg                                # :475-477
        if not (lost or enough):                                                  # :479
            continue
        Wp, X = intersection_of_lines(tracks.line_base[a:b], tracks.line_dir[a:b], tracks.line_conf[a:b])
        world[j] = Wp
        sv = np.linalg.svd(X, compute_uv=False)
        cond[j] = sv[0] / max(sv[-1], 1e-300)
        s = int(prob.obs_slot[a])                                                 # camera_indices[0], :481
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = prob.cam_R[s], prob.cam_t[s]
        Ti = np.linalg.inv(T)                                                     # geometry.py:35-37
        Cp = Ti[:3, :3] @ Wp + Ti[:3, 3]                                          # Camera.py:46-52
        flags[j] = FLAG_VALID | (FLAG_LOST if lost else 0)                        # :492-493
        if Cp[2] <= 0:                                                            # Camera.py:18
            continue
        im = K @ Cp
        im = im[:2] / im[2]                                                       # Camera.py:20-21
        if im[0] < 0 or im[0] >= params.width or im[1] < 0 or im[1] >= params.height:   # :24-26
            continue
        Wv = prob.cam_R[s] @ (Kinv @ np.append(im, 1.0))                          # MSCKF.py:486-487
        idp_rho[j] = 1.0 / Cp[2]                                                  # geometry.py:61-62
        theta = np.arctan2(Wv[0], Wv[2])                                          # geometry.py:64-67
        phi = np.arctan2(-Wv[1], np.sqrt(Wv[0] ** 2 + Wv[2] ** 2))
        idp_m[j] = [np.cos(phi) * np.sin(theta), -np.sin(phi), np.cos(phi) * np.cos(theta)]
        flags[j] |= FLAG_REFRESHED
    return dict(flags=flags, idp_m=idp_m, idp_rho=idp_rho, world=world, cond=cond)


# ---- f4: the geometric consistency tests of add_camera_measurements ----------------------------
def associate(prob, matched_uv, R_cur, t_cur, K, thr_epipolar=5.0, thr_homography=5.0):
    """The per-(match, earlier view) loop of `MSCKF.add_camera_measurements` (`MSCKF.py:345-412`) on the flat
    arrays: result 0 kept / 1 epipolar failure (:392-397) / 2 homography failure (:368-379) / 3 no match, and
    the index of the view that failed (-1)."""
    K = np.asarray(K, dtype=np.float64)
    invK = np.linalg.inv(K)                                      # :345
    F = prob.F
    res = np.zeros(F, dtype=np.uint8)
    fail = -np.ones(F, dtype=np.int32)
    T2 = np.eye(4); T2[:3, :3], T2[:3, 3] = R_cur, t_cur
    for i in range(F):
        mk = np.asarray(matched_uv[i], dtype=np.float64)
        if np.isnan(mk).any():
            res[i] = 3
            continue
        a, b = int(prob.view_ptr[i]), int(prob.view_ptr[i + 1])
        for j in range(a, b):                                    # :361
            fk = prob.obs_uv[j]
            s = int(prob.obs_slot[j])
            T1 = np.eye(4); T1[:3, :3], T1[:3, 3] = prob.cam_R[s], prob.cam_t[s]
            T12 = np.linalg.inv(T1) @ T2                         # :366
            R12, t12 = T12[:3, :3], T12[:3, 3]
            if np.linalg.norm(t12) < 0.01:                       # :368
                H = K @ R12 @ invK
                x1 = np.linalg.inv(H) @ np.array([mk[0], mk[1], 1.0])
                x1 = x1[:2] / x1[2]
                x2 = H @ np.array([fk[0], fk[1], 1.0])
                x2 = x2[:2] / x2[2]
                score = (np.linalg.norm(mk - x1) + np.linalg.norm(fk - x2)) / 2      # :374
                if score > thr_homography:
                    res[i], fail[i] = 2, j - a
                    break
            else:
                Fm = invK.T @ skew(t12) @ R12 @ invK             # :392
                score = np.append(mk, 1.0) @ Fm @ np.append(fk, 1.0)                  # :393 (signed)
                if score > thr_epipolar:
                    res[i], fail[i] = 1, j - a
                    break
    return res, fail


# ---- f2 / f3: the covariance steps either side of update -------------------------------------
def imu_transition(R, t, v, R0, t0, v0, gyro, acc, dt, gravity, w_planet, Qc):
    """Phi (15x15) and the discrete noise Q of `MSCKF.process_imu` (`MSCKF.py:179-237`).
    R, t, v are the IMU pose/velocity AFTER `imu.integrate` (:168), R0/t0/v0 the null state,
    gyro/acc the bias-corrected measurement (:166-167)."""
    F = np.zeros((15, 15))
    F[0:3, 0:3] = -skew(gyro)                                                  # :182
    F[0:3, 3:6] = -np.eye(3)                                                   # :183
    F[6:9, 0:3] = -R @ skew(acc)                                               # :186
    F[6:9, 6:9] = -2 * skew(w_planet)                                          # :187
    F[6:9, 9:12] = -R                                                          # :188
    F[6:9, 12:15] = -skew(w_planet) @ -skew(w_planet)                          # :189
    F[12:15, 6:9] = np.eye(3)                                                  # :192
    G = np.zeros((15, 12))
    G[0:3, 0:3] = -np.eye(3)                                                   # :203
    G[3:6, 3:6] = np.eye(3)                                                    # :206
    G[6:9, 6:9] = -R                                                           # :209
    G[9:12, 9:12] = np.eye(3)                                                  # :212
    Fdt = F * dt
    Fdt2 = Fdt @ Fdt
    Fdt3 = Fdt2 @ Fdt
    Phi = np.eye(15) + Fdt + 0.5 * Fdt2 + (1.0 / 6.0) * Fdt3                   # :218
    Phi[:3, :3] = R @ R0.T                                                     # :221
    u = R0 @ gravity                                                           # :223
    sv = u / (u @ u)                                                           # :224
    A_vel = Phi[6:9, :3].copy()
    A_pos = Phi[12:15, :3].copy()
    w1 = skew(v0 - v) @ gravity                                                # :229
    w2 = skew(dt * v0 + t0 - t) @ gravity                                      # :230
    Phi[6:9, :3] = A_vel - (A_vel @ u - w1)[:, None] * sv                      # :232
    Phi[12:15, :3] = A_pos - (A_pos @ u - w2)[:, None] * sv                    # :233
    Q = Phi @ G @ Qc @ G.T @ Phi.T * dt                                        # :237
    return Phi, Q


def propagate_covariance(P, Phi, Q):
    """`MSCKF.py:236-244`."""
    P = P.copy()
    P[:15, :15] = Phi @ P[:15, :15] @ Phi.T + Q
    P[:15, 15:] = Phi @ P[:15, 15:]
    P[15:, :15] = P[:15, 15:].T
    return (P + P.T) / 2


def augmentation_jacobian(imu_R, imu_t, T_W_I_R, T_W_I_t, T_W_C_R, T_W_C_t):
    """New clone pose and the 6x15 non-zero part of J, `MSCKF.py:252-261`
    (Isometry3D product/inverse through 4x4 matrices, `geometry.py:31-37`)."""
    def mat(R, t):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        return T
    T_I_C = np.linalg.inv(mat(T_W_I_R, T_W_I_t)) @ mat(T_W_C_R, T_W_C_t)       # :252
    T_W_Ci = mat(imu_R, imu_t) @ T_I_C                                         # :253
    J = np.zeros((6, 15))
    J[:3, :3] = T_I_C[:3, :3].T                                                # :259
    J[3:6, :3] = skew(imu_R @ T_I_C[:3, 3])                                    # :260
    J[3:6, 12:15] = np.eye(3)                                                  # :261
    return J, T_W_Ci[:3, :3], T_W_Ci[:3, 3]


def augment_covariance(P, J15):
    """`MSCKF.py:258-265`."""
    d = P.shape[0]
    J = np.zeros((6, d))
    J[:, :15] = J15
    M = np.vstack((np.eye(d), J))
    S = M @ P @ M.T
    return (S + S.T) / 2


def remove_clones_covariance(P, slots):
    """`MSCKF.remove_cameras` (`MSCKF.py:751-758`); slots are positions in the clone order
    at call time."""
    keep = np.ones(P.shape[0], dtype=bool)
    for s in slots:
        keep[15 + 6 * s:15 + 6 * (s + 1)] = False
    return P[np.ix_(keep, keep)]


def so3_correction(R, dtheta):
    """R <- R Exp(dtheta)^T followed by the SVD clean-up.
    reference `MSCKF.py:625-635` (IMU) and `:649-660` (clones)."""
    n = np.linalg.norm(dtheta)
    S = skew(dtheta)
    if np.isclose(n, 0):
        E = np.eye(3)
    else:
        E = np.eye(3) + (np.sin(n) / n) * S + ((1 - np.cos(n)) / n ** 2) * (S @ S)
    Rn = R @ E.T
    U, _, Vt = np.linalg.svd(Rn)
    return U @ Vt


def inject(dx, imu_R, imu_t, imu_v, imu_bg, imu_ba, cam_R, cam_t):
    """State injection half of `MSCKF.correct` (reference `MSCKF.py:616-661`).
    Returns new copies (imu_R, imu_t, imu_v, imu_bg, imu_ba, cam_R, cam_t)."""
    imu_R = so3_correction(imu_R, dx[0:3])
    imu_bg = imu_bg + dx[3:6]
    imu_v = imu_v + dx[6:9]
    imu_ba = imu_ba + dx[9:12]
    imu_t = imu_t + dx[12:15]
    cam_R = cam_R.copy()
    cam_t = cam_t.copy()
    for i in range(cam_R.shape[0]):
        dc = dx[15 + 6 * i:21 + 6 * i]
        cam_R[i] = so3_correction(cam_R[i], dc[:3])
        cam_t[i] = cam_t[i] + dc[3:6]
    return imu_R, imu_t, imu_v, imu_bg, imu_ba, cam_R, cam_t


def invariants(prob, accepted=None):
    """Basis-invariant per-problem quantities used to check intermediate HIP
    stages: G = H_X^T H_X (d, d) and b = H_X^T r_o (d,) over the accepted set."""
    d = prob.P.shape[0]
    G = np.zeros((d, d))
    b = np.zeros(d)
    for j in range(prob.F):
        if accepted is not None and not accepted[j]:
            continue
        r, H_x, H_f = feature_blocks(prob, j)
        r_o, H_o = project_on_nullspace(H_f, r, H_x)
        G += H_o.T @ H_o
        b += H_o.T @ r_o
    return G, b
