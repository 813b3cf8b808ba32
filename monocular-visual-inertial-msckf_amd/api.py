"""Host-side mirror of the reference's update interface.

`UpdateEngine.update_problem` is the flat call; `UpdateEngine.update` accepts
reference-shaped objects (an ordered `cameras` mapping, `Feature`-like records
with `keypoints`, `camera_indices`, `inverse_depth_point.{base,m,rho}`) exactly
like `MSCKF.update(features)` (reference `src/msckf/MSCKF.py:570`) reads them, and
then applies the state injection half of `MSCKF.correct` (`:616-661`) on the
host.  All arithmetic of the update itself runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _ffi
from .synth import SelectParams, TrackTable, UpdateProblem

_CHI2 = None


def chi2_table() -> np.ndarray:
    """chi2.ppf(0.95, dof) for dof = 0..512 (index 0 unused), generated once with
    scipy by tests/golden/gen_golden.py (reference `MSCKF.py:565-566`)."""
    global _CHI2
    if _CHI2 is None:
        _CHI2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "chi2_ppf_095.npy"))
    return _CHI2


@dataclass
class UpdateResult:
    status: int              # 0 updated, 1 no-op (nothing accepted)
    dx: np.ndarray           # (d,)
    P_new: np.ndarray        # (d, d)
    accepted: np.ndarray     # (F,) uint8, input order
    stats: dict

    @property
    def n_rejected(self) -> int:
        """Features that went through the gate and failed it (unselected ones do not count)."""
        if "n_rejected" in self.stats:
            return int(self.stats["n_rejected"])
        return int(self.accepted.size - int(self.accepted.sum()))


@dataclass
class Selection:
    """Outcome of `get_valid_features` (reference `MSCKF.py:458-495`), input order."""
    flags: np.ndarray        # (F,) uint8: 1 valid, 2 lost, 4 inverse-depth point refreshed
    idp_m: np.ndarray        # (F, 3) after the refresh
    idp_rho: np.ndarray      # (F,)
    world: np.ndarray        # (F, 3) triangulated points, NaN where none

    @property
    def valid(self) -> np.ndarray:
        return (self.flags & _ffi.SEL_VALID) > 0

    @property
    def lost(self) -> np.ndarray:
        return (self.flags & _ffi.SEL_LOST) > 0

    @property
    def refreshed(self) -> np.ndarray:
        return (self.flags & _ffi.SEL_REFRESHED) > 0


class UpdateEngine:
    """One context on one MI355X.  Not thread-safe (one engine per host thread)."""

    def __init__(self, max_clones: int = 30, max_features: int = 4096, max_track: int = 30,
                 device: int = 0, leaf_rows: int = 0, merge_arity: int = 0, plan: str = "auto", dtype: str = "f64"):
        """plan: "auto" = band pipeline (k_sweep) for tracks of up to 10 clone slots; longer ones are split into <= 10-slot
        blocks for that pipeline + a few remainder rows; "band" = no split: 90-column band tiles for tracks of
        11 - 15 slots, the merge tree beyond; "tree" = always the merge tree (A/B, tests).
        dtype: "f64" = the reference's arithmetic (parity 1e-8); "f32" = fp32 storage of the stacked system and
        the Joseph covariance update on the f32 matrix cores (BASELINE.json configs[4]; tolerance in DESIGN.md)."""
        self._lib = _ffi.load()
        if max_track > _ffi.MAX_TRACK:
            raise ValueError(f"max_track {max_track} > {_ffi.MAX_TRACK}")
        if plan not in ("auto", "band", "tree"):
            raise ValueError("plan must be 'auto', 'band' or 'tree'")
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        self.dtype = dtype
        cfg = _ffi.Config(_ffi.ABI_VERSION, device, max_clones, max_features, max_track, leaf_rows, merge_arity,
                          {"tree": _ffi.FLAG_TREE_PLAN, "band": _ffi.FLAG_BAND_ONLY}.get(plan, 0), _ffi.DTYPE_F32 if dtype == "f32" else _ffi.DTYPE_F64, 0)
        h = C.c_void_p()
        rc = self._lib.msckf_create(C.byref(h), C.byref(cfg))
        if rc != 0:
            raise _ffi.EngineError(rc, self._lib.msckf_strerror(rc).decode())
        self._h = h
        self.max_clones, self.max_features, self.max_track = max_clones, max_features, max_track
        self._N = 0
        self._F = 0

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.msckf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, allow_noop=True):
        if rc == 0 or (allow_noop and rc == 1):
            return rc
        text = self._lib.msckf_strerror(rc).decode()
        if rc in (_ffi.ERR_HIP, _ffi.ERR_COMM):
            text += ": " + self._lib.msckf_last_error(self._h).decode()
        raise _ffi.EngineError(rc, text)

    # -- one-shot drop-in ---------------------------------------------------
    def update_problem(self, prob: UpdateProblem) -> UpdateResult:
        """Host arrays in, host arrays out: the arithmetic of `MSCKF.update` +
        the covariance update of `MSCKF.correct` (reference `MSCKF.py:570-614`)."""
        N, F, d = prob.N, prob.F, prob.d
        a = self._pack(prob)
        chi = _ffi.f64(chi2_table())
        dx = np.empty(d)                           # the library writes every entry of dx, P_out, accepted[:F]
        P_out = np.empty((d, d))
        acc = np.zeros(max(F, 1), dtype=np.uint8)
        st = _ffi.Stats()
        rc = self._lib.msckf_update(
            self._h, N, _ffi.dptr(a["P"]), _ffi.dptr(a["cam_R"]), _ffi.dptr(a["cam_t"]), _ffi.dptr(a["cam_R0"]),
            _ffi.dptr(a["cam_t0"]), _ffi.dptr(a["g"]), _ffi.dptr(a["Kinv"]), float(prob.sigma), F,
            _ffi.iptr(a["view_ptr"]), _ffi.dptr(a["obs_uv"]), _ffi.iptr(a["obs_slot"]), _ffi.dptr(a["idp_base"]),
            _ffi.dptr(a["idp_m"]), _ffi.dptr(a["idp_rho"]), _ffi.dptr(chi), int(chi.size),
            _ffi.dptr(dx), _ffi.dptr(P_out), _ffi.uptr(acc), C.byref(st))
        self._check(rc)
        self._N, self._F = N, F
        return UpdateResult(rc, dx, P_out, acc[:F].copy(), st.as_dict())

    _kinv_cache = (None, None)

    @classmethod
    def _kinv(cls, K) -> np.ndarray:
        """inv(K) as the reference forms it (`MSCKF.py:519`); the intrinsics rarely change between calls."""
        K = np.asarray(K, dtype=np.float64)
        key, val = cls._kinv_cache
        if key is None or key.shape != K.shape or not np.array_equal(key, K):
            val = _ffi.f64(np.linalg.inv(K))
            cls._kinv_cache = (K.copy(), val)
        return val

    @classmethod
    def _pack(cls, prob: UpdateProblem) -> dict:
        # (C-contiguous float64 / int32 arrays whatever their shape: the library takes addresses)
        return dict(
            P=_ffi.f64(prob.P), cam_R=_ffi.f64(prob.cam_R), cam_t=_ffi.f64(prob.cam_t),
            cam_R0=_ffi.f64(prob.cam_R0), cam_t0=_ffi.f64(prob.cam_t0),
            g=_ffi.f64(prob.gravity), Kinv=cls._kinv(prob.K),               # reference MSCKF.py:519
            view_ptr=_ffi.i32(prob.view_ptr), obs_uv=_ffi.f64(prob.obs_uv),
            obs_slot=_ffi.i32(prob.obs_slot), idp_base=_ffi.f64(prob.idp_base),
            idp_m=_ffi.f64(prob.idp_m), idp_rho=_ffi.f64(prob.idp_rho))

    # -- resident path ------------------------------------------------------
    def set_state(self, prob: UpdateProblem):
        a = self._pack(prob)
        chi = _ffi.f64(chi2_table())
        self._check(self._lib.msckf_set_state(
            self._h, prob.N, _ffi.dptr(a["P"]), _ffi.dptr(a["cam_R"]), _ffi.dptr(a["cam_t"]), _ffi.dptr(a["cam_R0"]),
            _ffi.dptr(a["cam_t0"]), _ffi.dptr(a["g"]), _ffi.dptr(a["Kinv"]), float(prob.sigma), _ffi.dptr(chi),
            int(chi.size)), allow_noop=False)
        self._N = prob.N

    def set_features(self, prob: UpdateProblem):
        a = self._pack(prob)
        self._check(self._lib.msckf_set_features(
            self._h, prob.F, _ffi.iptr(a["view_ptr"]), _ffi.dptr(a["obs_uv"]), _ffi.iptr(a["obs_slot"]),
            _ffi.dptr(a["idp_base"]), _ffi.dptr(a["idp_m"]), _ffi.dptr(a["idp_rho"])), allow_noop=False)
        self._F = prob.F

    def load(self, prob: UpdateProblem):
        self.set_state(prob)
        self.set_features(prob)

    def run(self):
        self._check(self._lib.msckf_run(self._h), allow_noop=False)

    def run_compress(self):
        self._check(self._lib.msckf_run_compress(self._h), allow_noop=False)

    def sync(self):
        self._check(self._lib.msckf_sync(self._h), allow_noop=False)

    def run_timed(self, iters: int, stages: bool = False):
        """`iters` back-to-back device pipelines timed with HIP events on the
        engine's stream.  Returns (ms_total, [us_feature, us_qr, us_gain] or None)."""
        ms = C.c_float(0)
        st = (C.c_float * 3)()
        self._check(self._lib.msckf_run_timed(self._h, iters, C.byref(ms), st if stages else None), allow_noop=False)
        return float(ms.value), ([float(x) for x in st] if stages else None)

    def result(self) -> UpdateResult:
        d = 15 + 6 * self._N
        dx = np.zeros(d)
        P_out = np.zeros((d, d))
        acc = np.zeros(max(self._F, 1), dtype=np.uint8)
        st = _ffi.Stats()
        rc = self._check(self._lib.msckf_get_result(self._h, _ffi.dptr(dx), _ffi.dptr(P_out), _ffi.uptr(acc), C.byref(st)))
        return UpdateResult(rc, dx, P_out, acc[:self._F].copy(), st.as_dict())

    def commit_covariance(self) -> int:
        return self._check(self._lib.msckf_commit_covariance(self._h))

    # -- f1: selection + triangulation in front of the update -----------------
    def set_tracks(self, tracks: TrackTable):
        """Lines and frame counters of the batch given to `set_features` (same order)."""
        b, dvec, c = _ffi.f64(tracks.line_base).reshape(-1), _ffi.f64(tracks.line_dir).reshape(-1), _ffi.f64(tracks.line_conf)
        lo, tr = _ffi.i32(tracks.lost_for), _ffi.i32(tracks.tracked_for)
        self._check(self._lib.msckf_set_tracks(self._h, _ffi.dptr(b), _ffi.dptr(dvec), _ffi.dptr(c), _ffi.iptr(lo),
                                               _ffi.iptr(tr)), allow_noop=False)

    def run_select(self, params: SelectParams, K):
        """Enqueue `get_valid_features` on the device; the following `run()` processes only
        the valid features, with their inverse-depth points refreshed in HBM."""
        sp = _ffi.SelectParamsC()
        sp.min_frames_lost, sp.min_frames_tracked = int(params.min_frames_lost), int(params.min_frames_tracked)
        sp.use_parallax, sp.width, sp.height = int(bool(params.use_parallax)), int(params.width), int(params.height)
        sp.min_parallax_deg = float(params.min_parallax_deg)
        sp.K = (C.c_double * 9)(*np.asarray(K, dtype=np.float64).reshape(9))
        self._check(self._lib.msckf_run_select(self._h, C.byref(sp)), allow_noop=False)

    def replan(self):
        """Plan the QR tree over the valid features only (syncs; optional after `run_select`)."""
        self._check(self._lib.msckf_replan(self._h), allow_noop=False)

    def clear_selection(self):
        self._check(self._lib.msckf_clear_selection(self._h), allow_noop=False)

    def selection(self) -> Selection:
        F = self._F
        fl = np.zeros(max(F, 1), dtype=np.uint8)
        m, rho, w = np.zeros((max(F, 1), 3)), np.zeros(max(F, 1)), np.zeros((max(F, 1), 3))
        self._check(self._lib.msckf_get_selection(self._h, _ffi.uptr(fl), _ffi.dptr(m), _ffi.dptr(rho), _ffi.dptr(w)),
                    allow_noop=False)
        return Selection(fl[:F].copy(), m[:F].copy(), rho[:F].copy(), w[:F].copy())

    def time_select(self, iters: int = 50) -> float:
        """Average device microseconds of the selection kernel (HIP events)."""
        us = C.c_float(0)
        self._check(self._lib.msckf_debug_time_select(self._h, iters, C.byref(us)), allow_noop=False)
        return float(us.value)

    def select_problem(self, prob: UpdateProblem, tracks: TrackTable, params: SelectParams) -> Selection:
        """Flat `get_valid_features`: upload, select, download."""
        self.load(prob)
        self.set_tracks(tracks)
        self.run_select(params, prob.K)
        return self.selection()

    # -- f4: geometric consistency tests of the front end's matches ------------------
    def associate(self, matched_uv, R_cur, t_cur, K, epipolar_threshold: float = 5.0, homography_threshold: float = 5.0):
        """The per-(match, earlier view) tests of `add_camera_measurements` (reference `MSCKF.py:332-412`) for the
        loaded batch (the tracks BEFORE the new view) against one matched keypoint each (`matched_uv` (F, 2) pixels,
        NaN rows = no match).  Returns (result (F,) uint8: 0 kept, 1 epipolar failure, 2 homography failure,
        3 no match; fail_view (F,) int32)."""
        F = self._F
        ap = _ffi.AssocParamsC()
        ap.K = (C.c_double * 9)(*np.asarray(K, dtype=np.float64).reshape(9))
        ap.R_cur = (C.c_double * 9)(*np.asarray(R_cur, dtype=np.float64).reshape(9))
        ap.t_cur = (C.c_double * 3)(*np.asarray(t_cur, dtype=np.float64).reshape(3))
        ap.epipolar_threshold, ap.homography_threshold = float(epipolar_threshold), float(homography_threshold)
        uv = _ffi.f64(matched_uv).reshape(-1)
        if uv.size != 2 * F:
            raise ValueError("matched_uv must be (F, 2)")
        res = np.zeros(max(F, 1), dtype=np.uint8)
        fv = np.zeros(max(F, 1), dtype=np.int32)
        self._check(self._lib.msckf_run_associate(self._h, C.byref(ap), _ffi.dptr(uv), _ffi.uptr(res), _ffi.iptr(fv)),
                    allow_noop=False)
        return res[:F].copy(), fv[:F].copy()

    # -- f2 / f3: covariance resident in HBM between frames ---------------------
    def set_prior(self, P, gravity, K, sigma, cam_R=None, cam_t=None, cam_R0=None, cam_t0=None):
        """Upload the filter state once; N = 0 (the 15x15 IMU prior, no clone yet) is allowed."""
        P = _ffi.f64(P)
        N = (P.shape[0] - 15) // 6
        z9, z3 = np.zeros((0, 3, 3)), np.zeros((0, 3))
        cam_R = z9 if cam_R is None else cam_R
        cam_t = z3 if cam_t is None else cam_t
        prob = UpdateProblem(P=P, cam_R=np.asarray(cam_R, dtype=np.float64).reshape(N, 3, 3),
                             cam_t=np.asarray(cam_t, dtype=np.float64).reshape(N, 3),
                             cam_R0=np.asarray(cam_R if cam_R0 is None else cam_R0, dtype=np.float64).reshape(N, 3, 3),
                             cam_t0=np.asarray(cam_t if cam_t0 is None else cam_t0, dtype=np.float64).reshape(N, 3),
                             gravity=np.asarray(gravity, dtype=np.float64), K=np.asarray(K), sigma=float(sigma),
                             view_ptr=np.zeros(1, dtype=np.int32), obs_uv=np.zeros((0, 2)),
                             obs_slot=np.zeros(0, dtype=np.int32), idp_base=z3, idp_m=z3, idp_rho=np.zeros(0))
        self.set_state(prob)
        self._F = 0

    def propagate(self, Phi, Q):
        """Covariance half of `process_imu` (reference `MSCKF.py:236-244`); async."""
        a, b = _ffi.f64(Phi).reshape(-1), _ffi.f64(Q).reshape(-1)
        if a.size != 225 or b.size != 225:
            raise ValueError("Phi and Q are 15x15")
        self._check(self._lib.msckf_propagate(self._h, _ffi.dptr(a), _ffi.dptr(b)), allow_noop=False)

    def augment(self, J15, R, t):
        """`state_augmentation` (reference `MSCKF.py:250-265`): one more clone, P grows by 6."""
        j, r, tt = _ffi.f64(J15).reshape(-1), _ffi.f64(R).reshape(-1), _ffi.f64(t).reshape(-1)
        if j.size != 90 or r.size != 9 or tt.size != 3:
            raise ValueError("J15 is 6x15, R 3x3, t 3")
        self._check(self._lib.msckf_augment(self._h, _ffi.dptr(j), _ffi.dptr(r), _ffi.dptr(tt)), allow_noop=False)
        self._N += 1
        self._F = 0

    def remove_clones(self, slots):
        """Covariance half of `remove_cameras` (reference `MSCKF.py:751-757`)."""
        s = _ffi.i32(np.asarray(slots).reshape(-1))
        self._check(self._lib.msckf_remove_clones(self._h, int(s.size), _ffi.iptr(s)), allow_noop=False)
        self._N -= int(s.size)
        self._F = 0

    def set_poses(self, cam_R, cam_t, cam_R0=None, cam_t0=None):
        """Clone poses after the host's state injection (reference `MSCKF.py:642-661`)."""
        r, t = _ffi.f64(cam_R).reshape(-1), _ffi.f64(cam_t).reshape(-1)
        r0 = r if cam_R0 is None else _ffi.f64(cam_R0).reshape(-1)
        t0 = t if cam_t0 is None else _ffi.f64(cam_t0).reshape(-1)
        if r.size != 9 * self._N or t.size != 3 * self._N or r0.size != r.size or t0.size != t.size:
            raise ValueError("pose arrays do not match the number of clones")
        self._check(self._lib.msckf_set_poses(self._h, _ffi.dptr(r), _ffi.dptr(t), _ffi.dptr(r0), _ffi.dptr(t0)),
                    allow_noop=False)

    def covariance(self) -> np.ndarray:
        """The resident prior covariance (d x d)."""
        n = C.c_int32(0)
        self._check(self._lib.msckf_get_covariance(self._h, None, C.byref(n)), allow_noop=False)
        d = 15 + 6 * int(n.value)
        P = np.zeros((d, d))
        self._check(self._lib.msckf_get_covariance(self._h, _ffi.dptr(P), C.byref(n)), allow_noop=False)
        return P

    @property
    def n_clones(self) -> int:
        return self._N

    # -- sharded path -------------------------------------------------------
    def block_doubles(self) -> int:
        return int(self._lib.msckf_block_doubles(self._h))

    def export_block(self, dst_ptr: Optional[int] = None):
        """Copy the local compressed block [R | Q^T r].  With `dst_ptr` (a device
        address) the copy stays in HBM; otherwise a host array is returned."""
        n_acc = C.c_int32(0)
        if dst_ptr is not None:
            self._check(self._lib.msckf_export_block(self._h, C.c_void_p(dst_ptr), 1, C.byref(n_acc)), allow_noop=False)
            return None, int(n_acc.value)
        dc = 6 * self._N
        blk = np.zeros((dc, dc + 1))
        self._check(self._lib.msckf_export_block(self._h, blk.ctypes.data_as(C.c_void_p), 0, C.byref(n_acc)),
                    allow_noop=False)
        return blk, int(n_acc.value)

    def merge_gain(self, blocks, total_accepted: int, n_blocks: Optional[int] = None):
        """Root side: QR-merge the gathered blocks and run K6-K7.  `blocks` is a
        host array (G, 6N, 6N+1) or an int device address (then pass n_blocks)."""
        if isinstance(blocks, (int, np.integer)):
            self._check(self._lib.msckf_run_merge_gain(self._h, C.c_void_p(int(blocks)), int(n_blocks), 1,
                                                       int(total_accepted)), allow_noop=False)
        else:
            b = _ffi.f64(blocks)
            self._check(self._lib.msckf_run_merge_gain(self._h, b.ctypes.data_as(C.c_void_p), int(b.shape[0]), 0,
                                                       int(total_accepted)), allow_noop=False)

    # -- group exchange: the sharded band pipeline ------------------------------
    @staticmethod
    def max_span(prob: UpdateProblem) -> int:
        """Longest track of the batch in clone slots (last slot - first slot + 1)."""
        if prob.F == 0:
            return 0
        vp = np.asarray(prob.view_ptr)
        slots = np.asarray(prob.obs_slot).reshape(-1)
        lo = np.minimum.reduceat(slots, vp[:-1])
        hi = np.maximum.reduceat(slots, vp[:-1])
        return int((hi - lo + 1).max())

    def band_ok(self, prob: UpdateProblem) -> bool:
        """True when every shard of `prob` is planned as the band pipeline ON THIS ENGINE: the library's own
        rule (`msckf_band_rule`: plan flag, sweep tile width, LDS budget), asked with the whole batch's longest
        track, so that all ranks agree before sharding.  Then the ranks may exchange group triangles
        (`export_groups` / `merge_groups`); otherwise root blocks (`export_block` / `merge_gain`)."""
        if prob.F == 0:
            return False
        rc = self._lib.msckf_band_rule(self._h, int(prob.N), self.max_span(prob))
        if rc < 0:
            self._check(rc, allow_noop=False)
        return rc >= 1

    def set_group_exchange(self, on: bool = True):
        """Plan the following batches with the group-record layout (call before `load`)."""
        self._check(self._lib.msckf_set_group_exchange(self._h, 1 if on else 0), allow_noop=False)

    def set_exchange_span(self, max_span: int):
        """Tell the engine the longest track of the WHOLE batch (clone slots) before its shard is loaded: every rank
        then lays its group record out for the sweep mode of the whole batch (also the ring-buffered modes, N > 37
        or tracks of 11 - 15 slots).  0 = not told (60-column k_sweep form only)."""
        self._check(self._lib.msckf_set_exchange_span(self._h, int(max_span)), allow_noop=False)

    def group_record_doubles(self) -> int:
        return int(self._lib.msckf_group_record_doubles(self._h))

    def export_groups(self, dst_ptr: Optional[int] = None, count: bool = True):
        """The shard's group triangles (one record, accepted count included) after `run_compress`; host array
        or HBM address.  `count=False` skips reading the gate results back (returns -1 for the count)."""
        n_acc = C.c_int32(-1)
        if dst_ptr is not None:
            self._check(self._lib.msckf_export_groups(self._h, C.c_void_p(dst_ptr), 1, C.byref(n_acc) if count else None),
                        allow_noop=False)
            return None, int(n_acc.value)
        rec = np.zeros(self.group_record_doubles())
        self._check(self._lib.msckf_export_groups(self._h, rec.ctypes.data_as(C.c_void_p), 0, C.byref(n_acc)),
                    allow_noop=False)
        return rec, int(n_acc.value)

    def merge_groups(self, records, total_accepted: int = -1, n_records: Optional[int] = None):
        """Root side: fold the shards' group triangles, one root sweep, K6-K7.  `records` is a host array
        (G, record_doubles) or an int device address (then pass n_records); `total_accepted` < 0 takes the
        sum of the counts the shards wrote into their records."""
        if isinstance(records, (int, np.integer)):
            self._check(self._lib.msckf_run_merge_groups(self._h, C.c_void_p(int(records)), int(n_records), 1,
                                                         int(total_accepted)), allow_noop=False)
        else:
            b = _ffi.f64(records)
            self._check(self._lib.msckf_run_merge_groups(self._h, b.ctypes.data_as(C.c_void_p), int(b.shape[0]), 0,
                                                         int(total_accepted)), allow_noop=False)

    def merge_groups_flags(self, records_ptr: int, n_records: int, flags: np.ndarray):
        """`merge_groups` without any device-to-host traffic: `flags` (n_records, N) uint8 says which first-slot
        groups each shard's record carries (the caller partitioned the batch, so it knows); the accepted counts
        are summed on the device.  `records_ptr` is an HBM address."""
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self._lib.msckf_run_merge_groups_flags(self._h, C.c_void_p(int(records_ptr)), int(n_records), 1,
                                                           fl.ctypes.data), allow_noop=False)

    # -- RCCL exchange behind the C-ABI (no PyTorch) -----------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(_ffi.COMM_ID_BYTES)
        self._check(self._lib.msckf_comm_unique_id(buf), allow_noop=False)
        return buf.raw

    def comm_init(self, rank: int, world: int, uid: bytes):
        if len(uid) != _ffi.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self._check(self._lib.msckf_comm_init(self._h, int(rank), int(world), C.c_char_p(uid)), allow_noop=False)

    def comm_destroy(self):
        self._check(self._lib.msckf_comm_destroy(self._h), allow_noop=False)

    def comm_buffer(self, n_doubles: int) -> int:
        """HBM scratch of the engine (address), e.g. the receive side of the gather on the merging rank."""
        p = self._lib.msckf_comm_buffer(self._h, int(n_doubles) * 8)
        if not p:
            raise _ffi.EngineError(_ffi.ERR_HIP, "comm buffer allocation failed")
        return int(p)

    def comm_gather(self, send_ptr: int, recv_ptr: int, count: int, root: int = 0):
        self._check(self._lib.msckf_comm_gather(self._h, C.c_void_p(int(send_ptr)), C.c_void_p(int(recv_ptr) if recv_ptr else None),
                                                int(count), int(root)), allow_noop=False)

    def comm_broadcast(self, ptr: int, count: int, root: int = 0):
        self._check(self._lib.msckf_comm_broadcast(self._h, C.c_void_p(int(ptr)), int(count), int(root)), allow_noop=False)

    def comm_allreduce(self, ptr: int, count: int, op: str = "sum"):
        self._check(self._lib.msckf_comm_allreduce(self._h, C.c_void_p(int(ptr)), int(count), 1 if op == "max" else 0),
                    allow_noop=False)

    def comm_put(self, dst_ptr: int, host: np.ndarray):
        a = _ffi.f64(host)
        self._check(self._lib.msckf_comm_put(self._h, C.c_void_p(int(dst_ptr)), a.ctypes.data, a.nbytes), allow_noop=False)

    def comm_get(self, src_ptr: int, n_doubles: int) -> np.ndarray:
        out = np.empty(int(n_doubles))
        self._check(self._lib.msckf_comm_get(self._h, out.ctypes.data, C.c_void_p(int(src_ptr)), out.nbytes), allow_noop=False)
        return out

    def result_host(self):
        """dx, P+ of the result range as host arrays without the gate bookkeeping (two copies: dx and P_out each
        from its own address)."""
        d = 15 + 6 * self._N
        dx = self.comm_get(self.device_pointer(0), d)
        P = self.comm_get(self.device_pointer(1), d * d)
        return dx, P.reshape(d, d)

    def device_pointer(self, which: int) -> int:
        """0 dx (dx | P_out contiguous for the current N), 1 P_out, 2 root block, 3 group record of the shard,
        4 prior covariance, 5 result range (status | dx | P_out | gate bytes), 6 its gate bytes."""
        return int(self._lib.msckf_device_pointer(self._h, int(which)))

    # -- sharded update: gate results riding with the exchange ---------------------------------
    def set_exchange_mask(self, bounds=None):
        """`bounds` (n_shards + 1,): shard r holds features [bounds[r], bounds[r+1]) of the whole batch; None
        switches the ride-along off.  Call before `load` / `set_features` (the record layout changes)."""
        if bounds is None:
            self._check(self._lib.msckf_set_exchange_mask(self._h, 0, None), allow_noop=False)
            self._F_total = 0
            return
        b = _ffi.i32(np.asarray(bounds).reshape(-1))
        self._check(self._lib.msckf_set_exchange_mask(self._h, int(b.size) - 1, _ffi.iptr(b)), allow_noop=False)
        self._F_total = int(b[-1])

    def gate_bytes(self) -> np.ndarray:
        """Gate byte per feature of the loaded batch, input order: 1 accepted, 0 otherwise (the group records
        carry the finer codes 2 = gate matrix not SPD, 3 = not selected; this host read-back, used by the
        root-block fallback exchange only, folds them into 0)."""
        st = _ffi.Stats()
        acc = np.zeros(max(self._F, 1), dtype=np.uint8)
        self._check(self._lib.msckf_get_result(self._h, None, None, _ffi.uptr(acc), C.byref(st)))
        return acc[:self._F]

    def result_range_doubles(self) -> int:
        return int(self._lib.msckf_result_range_doubles(self._h))

    def shared_result(self) -> UpdateResult:
        """The complete result of a sharded update as read from the result range (any rank, after the
        broadcast): status, dx, P+, accepted[F_total] of the whole batch in input order, gate counters."""
        d = 15 + 6 * self._N
        F = getattr(self, "_F_total", 0)
        dx, P_out = np.empty(d), np.empty((d, d))
        acc = np.zeros(max(F, 1), dtype=np.uint8)
        st = _ffi.Stats()
        rc = self._check(self._lib.msckf_get_shared_result(self._h, _ffi.dptr(dx), _ffi.dptr(P_out), _ffi.uptr(acc), C.byref(st)))
        return UpdateResult(rc, dx, P_out, acc[:F].copy(), st.as_dict())

    def result_device_view(self):
        """`dx | P+` of the last run as ONE contiguous HBM range (d + d*d doubles) exposed through
        `__cuda_array_interface__`: `torch.as_tensor(view, device="cuda")` wraps it without a copy, e.g. as the
        send buffer of the broadcast that follows the merge on rank 0.  Valid until the engine is closed; its
        contents change with every run (synchronise the engine's stream first)."""
        d = 15 + 6 * self._N
        ptr = int(self._lib.msckf_device_pointer(self._h, 0))

        class _View:
            __cuda_array_interface__ = {"shape": (d + d * d,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        return _View()

    def export_result(self, dx_ptr: int, P_ptr: int):
        """dx and P+ of the last run into HBM buffers owned by the caller."""
        self._check(self._lib.msckf_export_result(self._h, C.c_void_p(dx_ptr), C.c_void_p(P_ptr), 1), allow_noop=False)

    def import_covariance(self, P):
        """New prior covariance from a host array or an int device address."""
        if isinstance(P, (int, np.integer)):
            self._check(self._lib.msckf_import_covariance(self._h, C.c_void_p(int(P)), 1), allow_noop=False)
        else:
            a = _ffi.f64(P)
            self._check(self._lib.msckf_import_covariance(self._h, a.ctypes.data_as(C.c_void_p), 0), allow_noop=False)

    # -- introspection ------------------------------------------------------
    def debug_gate(self):
        g = np.zeros(max(self._F, 1))
        q = np.zeros(max(self._F, 1), dtype=np.int32)
        self._check(self._lib.msckf_debug_gate(self._h, _ffi.dptr(g), _ffi.iptr(q)), allow_noop=False)
        return g[:self._F], q[:self._F]

    def debug_split(self) -> dict:
        """How the loaded batch's long tracks were planned (`msckf_debug_split`)."""
        out = (C.c_int32 * 8)()
        self._check(self._lib.msckf_debug_split(self._h, out), allow_noop=False)
        names = ("long_tracks", "narrow_blocks", "remainder_rows_cap", "remainder_mode", "remainder_tree_levels", "entries",
                 "band_plan", "sweep_mode")
        return dict(zip(names, [int(x) for x in out]))

    def set_rem_direct_rows(self, rows: int = -1):
        """Tests: remainder rows up to which K6-K7 takes them without a QR of their own (< 0: the default)."""
        self._check(self._lib.msckf_debug_set_rem_direct_rows(self._h, int(rows)), allow_noop=False)

    def debug_compressed(self):
        dc = 6 * self._N
        T = np.zeros((dc, dc))
        rn = np.zeros(dc)
        self._check(self._lib.msckf_debug_compressed(self._h, _ffi.dptr(T), _ffi.dptr(rn)), allow_noop=False)
        return T, rn

    # -- reference-shaped drop-in ---------------------------------------------
    def update(self, filt, features) -> int:
        """Drop-in for `MSCKF.update(features)` (reference `MSCKF.py:570-609`)
        including `correct` (`:611-661`).  `filt` is any object with the
        attributes the reference method reads: `state.cameras` (ordered mapping
        key -> camera with `T_W_Ci`, `T_W_Ci_null`, each with `.R`, `.t`),
        `state.covariance`, `state.imu` (`W_gravity`, `T_W_Ii`, `v_W_Ii`,
        `gyroscope_bias`, `accelerometer_bias`), `K`, `sigma_image`,
        `number_of_residuals_discarded_for_gasting_test`.
        Returns the status (0 updated / 1 no-op); mutates `filt` like the reference."""
        from .pack import problem_from_reference
        from .inject import inject_state
        prob = problem_from_reference(filt, features)
        res = self.update_problem(prob)
        filt.number_of_residuals_discarded_for_gasting_test += res.n_rejected        # MSCKF.py:578
        if res.status != 0:
            return res.status                                                         # MSCKF.py:584-585
        filt.state.covariance = res.P_new                                             # MSCKF.py:614 (rebinds)
        inject_state(filt.state, res.dx)                                              # MSCKF.py:616-661
        return 0

    def process_features(self, filt) -> int:
        """Drop-in for `MSCKF.process_features()` (reference `MSCKF.py:450-456`):
        `get_valid_features(self.features)` and `update(valid_features)` in one device pass,
        then `remove_features(lost_features)` through the filter's own method.  Mutates the
        features' inverse-depth points (`:488`) and the filter like the reference does.
        Returns 0 updated / 1 no-op."""
        from .pack import problem_from_reference, select_params_from_reference, tracks_from_reference
        from .inject import inject_state
        feats = filt.features
        if len(feats) == 0:
            return 1
        prob = problem_from_reference(filt, feats)
        self.load(prob)
        self.set_tracks(tracks_from_reference(feats))
        self.run_select(select_params_from_reference(filt), prob.K)
        sel = self.selection()
        if 0 < int(sel.valid.sum()) < 0.15 * len(feats):
            self.replan()                      # few valid candidates: a shallower QR tree pays for the sync
        self.run()
        res = self.result()
        items = list(feats.items())
        for j, (_, ft) in enumerate(items):
            if sel.flags[j] & _ffi.SEL_REFRESHED:                                     # MSCKF.py:488
                ft.inverse_depth_point.m = sel.idp_m[j].copy()
                ft.inverse_depth_point.rho = float(sel.idp_rho[j])
                if hasattr(filt, "estimated_world_points"):
                    filt.estimated_world_points.append(sel.world[j].copy())           # :489
        if not sel.valid.any():
            return 1                                                                  # :454
        filt.number_of_residuals_discarded_for_gasting_test += res.n_rejected        # :578
        if res.status == 0:
            filt.state.covariance = res.P_new                                         # :614
            inject_state(filt.state, res.dx)                                          # :616-661
        filt.remove_features({k: ft for j, (k, ft) in enumerate(items) if sel.flags[j] & _ffi.SEL_LOST})   # :456
        return res.status

    def prune_poorest_camera_states(self, filt) -> int:
        """Drop-in for `MSCKF.prune_poorest_camera_states()` (reference `MSCKF.py:710-737`): the two clones seen by
        the fewest features, the features seen by them -> `get_valid_features` -> `update` -> `remove_cameras`,
        composed on the resident engine (`_prune`).  Returns 0 updated / 1 no update."""
        count = {}
        for ft in filt.features.values():                                             # :712-716
            for ci in ft.camera_indices:
                count[ci] = count.get(ci, 0) + 1
        poorest = [k for k, _ in sorted(count.items(), key=lambda kv: kv[1])][:2]     # :718-724 (stable sort, dict order)
        return self._prune(filt, {k: filt.state.cameras[k] for k in poorest})

    def prune_camera_states(self, filt) -> int:
        """Drop-in for `MSCKF.prune_camera_states()` (reference `MSCKF.py:663-680`): every
        `int(max_number_of_camera_states / camera_states_to_delete)`-th clone of the window (position i > 0 with
        i % step == 0, `:665-667`), the features seen by them -> `get_valid_features` -> `update` ->
        `remove_cameras`, composed on the resident engine (`_prune`).  Returns 0 updated / 1 no update."""
        step = int(filt.max_number_of_camera_states / filt.camera_states_to_delete)   # :666
        drop = {k: cam for i, (k, cam) in enumerate(filt.state.cameras.items()) if i > 0 and i % step == 0}
        return self._prune(filt, drop)

    def _prune(self, filt, drop) -> int:
        """Shared tail of the reference's two pruning methods (`MSCKF.py:669-680`, `:726-737`): selection, update,
        commit and the removal of the clones' rows / columns (`:751-757`) run back to back in HBM; the covariance
        crosses PCIe once in each direction.  The dictionary bookkeeping of `remove_cameras` (`:758-777`, including
        the `last_camera_measurement` entries of features that lost all their views) stays on the host."""
        from .pack import problem_from_reference, select_params_from_reference, tracks_from_reference
        from .inject import inject_state
        cams = filt.state.cameras
        todo = {i: ft for i, ft in filt.features.items() if any(ci in drop for ci in ft.camera_indices)}   # :669-674
        keys = list(cams.keys())
        slots = [keys.index(k) for k in drop]
        status = 1
        if len(todo) > 0:
            prob = problem_from_reference(filt, todo)
            self.load(prob)
            self.set_tracks(tracks_from_reference(todo))
            self.run_select(select_params_from_reference(filt), prob.K)
            sel = self.selection()
            items = list(todo.items())
            for j, (_, ft) in enumerate(items):
                if sel.flags[j] & _ffi.SEL_REFRESHED:                                 # :488
                    ft.inverse_depth_point.m = sel.idp_m[j].copy()
                    ft.inverse_depth_point.rho = float(sel.idp_rho[j])
                    if hasattr(filt, "estimated_world_points"):
                        filt.estimated_world_points.append(sel.world[j].copy())
            if sel.valid.any():                                                       # :676-678
                self.run()
                d = 15 + 6 * self._N
                dx = np.zeros(d)
                acc = np.zeros(max(self._F, 1), dtype=np.uint8)
                st = _ffi.Stats()
                status = self._check(self._lib.msckf_get_result(self._h, _ffi.dptr(dx), None, _ffi.uptr(acc), C.byref(st)))
                filt.number_of_residuals_discarded_for_gasting_test += int(st.n_rejected)     # :578
                if status == 0:
                    self.commit_covariance()                                          # :614, P+ stays in HBM
                    inject_state(filt.state, dx)                                      # :616-661 (before the clones go)
        else:
            self.set_prior(filt.state.covariance, filt.state.imu.W_gravity, filt.K, filt.sigma_image,
                           [c.T_W_Ci.R for c in cams.values()], [c.T_W_Ci.t for c in cams.values()])
        if slots:
            self.remove_clones(slots)                                                 # :751-757 on the resident P
        filt.state.covariance = self.covariance()
        for k in drop:                                                                # :758
            del cams[k]
        gone = []
        for i, ft in filt.features.items():                                           # :760-769
            for k in drop:
                if k in ft.camera_indices:
                    v = ft.camera_indices.index(k)
                    for name in ("keypoints", "descriptors", "scores", "camera_indices", "lines"):
                        lst = getattr(ft, name, None)
                        if lst is not None and len(lst) > v:
                            del lst[v]
            if len(ft.camera_indices) == 0:
                gone.append(i)
        lcm = getattr(filt, "last_camera_measurement", None)
        for i in gone:                                                                # :771-777
            del filt.features[i]
            if lcm is not None:
                idx = np.where(lcm.features_indices == i)[0]
                if len(idx) > 0:
                    lcm.descriptors = np.delete(lcm.descriptors, idx[0], axis=0)
                    lcm.features_indices = np.delete(lcm.features_indices, idx[0], axis=0)
        return status
