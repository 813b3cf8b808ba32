"""Seeded synthetic (N_clones, N_features, track_len) workloads for the MSCKF
measurement-update path.

NumPy only.  The recipe follows SURVEY.md §8(d) and borrows the camera
constants of the reference's defaults (reference `src/msckf/MSCKF.py:18-26`:
K = [[180,0,320],[0,180,240],[0,0,1]], 640x480, camera axes T_W_C, sigma 0.2,
gravity [0,0,-9.81]).  The output is the flat "update problem" the C-ABI takes
(see include/msckf_mi355x.h): covariance, clone poses (current and null), CSR
feature tracks and inverse-depth points.

The inverse-depth parametrisation reproduces the reference's own convention
(`src/utils/geometry.py:53-71`, `src/msckf/MSCKF.py:485-488`): `m` is a *unit*
bearing in the world frame built through (theta, phi), `rho` = 1 / z-depth in
the anchor camera.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

K_DEFAULT = np.array([[180.0, 0.0, 320.0], [0.0, 180.0, 240.0], [0.0, 0.0, 1.0]])
R_WC_DEFAULT = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
GRAVITY_DEFAULT = np.array([0.0, 0.0, -9.81])
WIDTH, HEIGHT = 640, 480


@dataclass
class UpdateProblem:
    """Flat description of one `MSCKF.update(features)` call (reference
    `src/msckf/MSCKF.py:570`), i.e. everything that call reads."""

    P: np.ndarray            # (d, d) covariance, d = 15 + 6 N
    cam_R: np.ndarray        # (N, 3, 3) R_W_Ci            (Camera.T_W_Ci.R)
    cam_t: np.ndarray        # (N, 3)    t_W_Ci            (Camera.T_W_Ci.t)
    cam_R0: np.ndarray       # (N, 3, 3) null-state R      (Camera.T_W_Ci_null.R)
    cam_t0: np.ndarray       # (N, 3)    null-state t      (Camera.T_W_Ci_null.t)
    gravity: np.ndarray      # (3,)      imu.W_gravity
    K: np.ndarray            # (3, 3)    pinhole matrix (self.K)
    sigma: float             # sigma_image
    view_ptr: np.ndarray     # (F+1,) int32 CSR offsets into obs_*
    obs_uv: np.ndarray       # (sum M, 2) pixel observations (Feature.keypoints)
    obs_slot: np.ndarray     # (sum M,) int32 clone slot = position in the ordered cameras dict
    idp_base: np.ndarray     # (F, 3) InverseDepthPoint.base
    idp_m: np.ndarray        # (F, 3) InverseDepthPoint.m
    idp_rho: np.ndarray      # (F,)   InverseDepthPoint.rho
    meta: dict = field(default_factory=dict)

    @property
    def N(self) -> int:
        return int(self.cam_R.shape[0])

    @property
    def F(self) -> int:
        return int(self.view_ptr.shape[0] - 1)

    @property
    def d(self) -> int:
        return 15 + 6 * self.N

    def take(self, idx) -> "UpdateProblem":
        """The features `idx` (any order, e.g. the ones `get_valid_features` selected) with the
        shared state kept; optional replacement inverse-depth points go in afterwards."""
        idx = np.asarray(idx, dtype=np.int64)
        lens = (self.view_ptr[1:] - self.view_ptr[:-1])[idx]
        vp = np.zeros(len(idx) + 1, dtype=np.int32)
        vp[1:] = np.cumsum(lens)
        rows = np.concatenate([np.arange(self.view_ptr[j], self.view_ptr[j + 1]) for j in idx]) if len(idx) \
            else np.zeros(0, dtype=np.int64)
        return UpdateProblem(
            P=self.P, cam_R=self.cam_R, cam_t=self.cam_t, cam_R0=self.cam_R0, cam_t0=self.cam_t0,
            gravity=self.gravity, K=self.K, sigma=self.sigma, view_ptr=vp,
            obs_uv=self.obs_uv[rows].reshape(-1, 2), obs_slot=self.obs_slot[rows].astype(np.int32),
            idp_base=self.idp_base[idx].reshape(-1, 3), idp_m=self.idp_m[idx].reshape(-1, 3),
            idp_rho=self.idp_rho[idx], meta=dict(self.meta, take=idx))

    def subset(self, lo: int, hi: int) -> "UpdateProblem":
        """Contiguous feature shard [lo, hi) with the shared state (P, poses) kept."""
        a, b = int(self.view_ptr[lo]), int(self.view_ptr[hi])
        return UpdateProblem(
            P=self.P, cam_R=self.cam_R, cam_t=self.cam_t, cam_R0=self.cam_R0, cam_t0=self.cam_t0,
            gravity=self.gravity, K=self.K, sigma=self.sigma,
            view_ptr=(self.view_ptr[lo:hi + 1] - a).astype(np.int32),
            obs_uv=self.obs_uv[a:b], obs_slot=self.obs_slot[a:b],
            idp_base=self.idp_base[lo:hi], idp_m=self.idp_m[lo:hi], idp_rho=self.idp_rho[lo:hi],
            meta=dict(self.meta, shard=(lo, hi)))


@dataclass
class SelectParams:
    """The `MSCKFParameters` fields `MSCKF.get_valid_features` reads (reference
    `src/msckf/MSCKF.py:24-25, 39-44` defaults)."""

    min_frames_lost: int = 1          # min_number_of_frames_to_be_lost
    min_frames_tracked: int = 5       # min_number_of_frames_to_be_tracked
    use_parallax: bool = True
    min_parallax_deg: float = 20.0
    width: int = WIDTH
    height: int = HEIGHT


@dataclass
class TrackTable:
    """Per-view `Feature.lines` and per-feature frame counters, aligned with an
    `UpdateProblem`'s CSR (`Feature.lines[i]` belongs to `Feature.keypoints[i]`,
    reference `MSCKF.py:305, 410, 430, 765-769`)."""

    line_base: np.ndarray     # (sum M, 3) Line.base  (camera position when the view was added)
    line_dir: np.ndarray      # (sum M, 3) Line.direction (world bearing, not normalised)
    line_conf: np.ndarray     # (sum M,)   Line.confidence (match score)
    lost_for: np.ndarray      # (F,) int32 Feature.lost_for_n_frames
    tracked_for: np.ndarray   # (F,) int32 Feature.tracked_for_n_frames


def make_tracks(prob: UpdateProblem, seed: int = 0, *, lost_fraction: float = 0.5,
                pose_jitter: float = 0.005, flip_fraction: float = 0.0) -> TrackTable:
    """Seeded lines and counters for `prob`'s tracks: every view's line starts at the
    clone position (plus a small jitter: the reference stores the pose the view was
    added with, `MSCKF.py:410`) along R K^-1 [u, v, 1]; `flip_fraction` of the features
    get their bearings in reversed view order, so that the lines cross behind the cameras."""
    rng = np.random.default_rng(5000 + seed)
    Kinv = np.linalg.inv(np.asarray(prob.K, dtype=np.float64))
    n = int(prob.view_ptr[-1])
    uv1 = np.concatenate([prob.obs_uv, np.ones((n, 1))], axis=1)
    dirs = np.einsum("nij,nj->ni", prob.cam_R[prob.obs_slot], uv1 @ Kinv.T)
    base = prob.cam_t[prob.obs_slot] + pose_jitter * rng.standard_normal((n, 3))
    conf = rng.uniform(0.2, 1.0, n)
    F = prob.F
    lens = prob.view_ptr[1:] - prob.view_ptr[:-1]
    lost = np.where(rng.uniform(size=F) < lost_fraction, rng.integers(1, 4, F), 0).astype(np.int32)
    tracked = np.maximum(1, lens + rng.integers(-2, 3, F)).astype(np.int32)    # views get pruned with their clones
    flip = rng.uniform(size=F) < flip_fraction
    for j in np.nonzero(flip)[0]:
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        dirs[a:b] = dirs[a:b][::-1].copy()
    return TrackTable(line_base=base, line_dir=dirs, line_conf=conf, lost_for=lost, tracked_for=tracked)


def so3_exp(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    W = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + (np.sin(th) / th) * W + ((1.0 - np.cos(th)) / th ** 2) * (W @ W)


def bearing_from_direction(direction: np.ndarray) -> np.ndarray:
    """Unit bearing through (theta, phi) exactly as the reference builds
    `InverseDepthPoint.m` (`src/utils/geometry.py:56-58, 64-67`)."""
    theta = np.arctan2(direction[0], direction[2])
    phi = np.arctan2(-direction[1], np.sqrt(direction[0] ** 2 + direction[2] ** 2))
    return np.array([np.cos(phi) * np.sin(theta), -np.sin(phi), np.cos(phi) * np.cos(theta)])


def random_spd_covariance(d: int, rng: np.random.Generator) -> np.ndarray:
    """Recipe A of SURVEY.md §8(c): P = 1e-4 A A^T / d + 1e-3 I (cond ~ 1.4)."""
    A = rng.standard_normal((d, d))
    P = 1e-4 * (A @ A.T) / d + 1e-3 * np.eye(d)
    return (P + P.T) / 2


def clone_poses(N: int, rng: np.random.Generator):
    R = np.empty((N, 3, 3))
    t = np.empty((N, 3))
    for i in range(N):
        t[i] = [0.15 * i, 0.05 * np.sin(0.3 * i), 0.02 * np.cos(0.2 * i)]
        R[i] = so3_exp(0.02 * rng.standard_normal(3)) @ R_WC_DEFAULT
    return R, t


def make_problem(N: int, F: int, M: int, seed: int = 0, *, sigma: float = 0.2,
                 pixel_noise: float = 0.2, outlier_fraction: float = 0.0,
                 outlier_px: float = 400.0, variable_tracks: bool = False,
                 min_track: int = 2, P: Optional[np.ndarray] = None,
                 distinct_null: bool = False, gravity: Optional[np.ndarray] = None,
                 K: Optional[np.ndarray] = None, poses=None) -> UpdateProblem:
    """Build one seeded update problem with N clones and F features tracked over
    M consecutive clones each (or a uniform length in [min_track, M] when
    `variable_tracks`)."""
    rng = np.random.default_rng(seed)
    K = K_DEFAULT.copy() if K is None else np.asarray(K)
    Kf = np.asarray(K, dtype=np.float64)
    Kinv = np.linalg.inv(Kf)
    g = GRAVITY_DEFAULT.copy() if gravity is None else np.asarray(gravity, dtype=np.float64)
    if poses is None:
        cam_R, cam_t = clone_poses(N, rng)
    else:
        cam_R, cam_t = np.array(poses[0], dtype=np.float64), np.array(poses[1], dtype=np.float64)
    d = 15 + 6 * N
    if P is None:
        P = random_spd_covariance(d, rng)
    if distinct_null:
        cam_R0 = np.stack([so3_exp(0.005 * rng.standard_normal(3)) @ cam_R[i] for i in range(N)])
        cam_t0 = cam_t + 0.01 * rng.standard_normal((N, 3))
    else:
        cam_R0, cam_t0 = cam_R.copy(), cam_t.copy()

    view_ptr = [0]
    obs_uv, obs_slot, bases, ms, rhos = [], [], [], [], []
    outlier_flags = []
    while len(rhos) < F:
        Mj = int(rng.integers(min_track, M + 1)) if variable_tracks else M
        s0 = int(rng.integers(0, N - Mj + 1))
        pc = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(4, 12)])
        pw = cam_R[s0] @ pc + cam_t[s0]
        uvs = []
        ok = True
        for v in range(Mj):
            s = s0 + v
            q = cam_R[s].T @ (pw - cam_t[s])
            if q[2] <= 0:
                ok = False
                break
            px = Kf @ q
            px = px[:2] / px[2]
            if not (0 <= px[0] < WIDTH and 0 <= px[1] < HEIGHT):
                ok = False
                break
            uvs.append(px + pixel_noise * rng.standard_normal(2))
        znoise = rng.standard_normal()
        is_outlier = rng.uniform() < outlier_fraction
        if not ok:
            continue
        if is_outlier:
            k = int(rng.integers(0, Mj))
            uvs[k] = uvs[k] + outlier_px * np.array([1.0, -0.7])
        ci_v = Kinv @ np.array([uvs[0][0], uvs[0][1], 1.0])
        w_v = cam_R[s0] @ ci_v
        bases.append(cam_t[s0].copy())
        ms.append(bearing_from_direction(w_v))
        rhos.append(1.0 / (pc[2] * (1.0 + 0.01 * znoise)))
        obs_uv.extend(uvs)
        obs_slot.extend(range(s0, s0 + Mj))
        view_ptr.append(len(obs_slot))
        outlier_flags.append(is_outlier)

    return UpdateProblem(
        P=P, cam_R=cam_R, cam_t=cam_t, cam_R0=cam_R0, cam_t0=cam_t0, gravity=g, K=K, sigma=float(sigma),
        view_ptr=np.asarray(view_ptr, dtype=np.int32),
        obs_uv=np.asarray(obs_uv, dtype=np.float64).reshape(-1, 2),
        obs_slot=np.asarray(obs_slot, dtype=np.int32),
        idp_base=np.asarray(bases, dtype=np.float64).reshape(-1, 3),
        idp_m=np.asarray(ms, dtype=np.float64).reshape(-1, 3),
        idp_rho=np.asarray(rhos, dtype=np.float64),
        meta={"N": N, "F": F, "M": M, "seed": seed, "outliers": np.asarray(outlier_flags)})


# The configurations BASELINE.json names (N clones, F features, track length).
CONFIGS = {
    "cfg1": (10, 50, 5),
    "cfg2": (20, 500, 8),
    "cfg3": (30, 2000, 10),      # headline
    "cfg4": (30, 8000, 10),      # feature-sharded, multi-GPU
    "cfg5": (50, 20000, 15),     # fp32 storage
    "north_star": (30, 10000, 10),
}


def concat_problems(a: UpdateProblem, b: UpdateProblem) -> UpdateProblem:
    """The tracks of `b` appended to those of `a` (same state: build `b` with P=a.P, poses=(a.cam_R, a.cam_t))."""
    vp = np.concatenate([a.view_ptr, a.view_ptr[-1] + b.view_ptr[1:]]).astype(np.int32)
    cat = lambda x, y: np.concatenate([x, y])
    return UpdateProblem(**{**a.__dict__, "view_ptr": vp, "obs_uv": cat(a.obs_uv, b.obs_uv), "obs_slot": cat(a.obs_slot, b.obs_slot),
                            "idp_base": cat(a.idp_base, b.idp_base), "idp_m": cat(a.idp_m, b.idp_m), "idp_rho": cat(a.idp_rho, b.idp_rho)})


def few_long_tracks_problem(N: int = 30, F: int = 2000, n_long: int = 10, M_short: int = 10, seed: int = 0) -> UpdateProblem:
    """F - n_long tracks of M_short views and n_long tracks that span the whole window: the shape a handful of long-lived
    features give a frame's batch (reference window: 30 clones, MSCKF.py:45; tracks grow a view per frame, :404-412)."""
    a = make_problem(N, F - n_long, M_short, seed=seed)
    b = make_problem(N, n_long, N, seed=seed + 100, P=a.P, poses=(a.cam_R, a.cam_t))
    return concat_problems(a, b)
