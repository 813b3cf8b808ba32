"""The two-level nullspace basis behind the split of long tracks, in NumPy against the oracle -- CPU only.

reference MSCKF.py:554-559 projects every track with scipy's basis of null(H_f^T); dx, P+ and the gate statistic do not depend on
which orthonormal basis is taken.  `k_feature<64, true>` builds one in two levels, so that most projected rows of a
long track touch at most 10 clone slots: this file restates that construction row for row (group-wise Householder, carry rows, a
second factorisation of the stacked carries) and checks it against the oracle's update."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from msckf_amd import synth
from oracle import msckf_oracle as oracle

GROUP_SLOTS = 10          # SPLIT_GSLOTS of csrc/k_feature.h


def householder_q(Hf):
    """Q (m x m) with Q^T Hf upper trapezoidal: min(m, 3) unpivoted reflectors (level 1 of the kernel)."""
    m = Hf.shape[0]
    Q, A = np.eye(m), Hf.copy()
    for k in range(min(3, m)):
        x = A[k:, k]
        nx = np.linalg.norm(x)
        if nx == 0.0:
            continue
        v = x.copy()
        v[0] += np.copysign(nx, x[0])
        v /= np.linalg.norm(v)
        Hk = np.eye(m)
        Hk[k:, k:] -= 2.0 * np.outer(v, v)
        A, Q = Hk @ A, Q @ Hk
    return Q


def groups_of(slots):
    """The kernel's view groups (msckf_set_features): ceil(span / 10) stretches of equal width, empty ones skipped."""
    lo, hi = int(slots.min()), int(slots.max())
    span = hi - lo + 1
    ng0 = (span + GROUP_SLOTS - 1) // GROUP_SLOTS
    out, v = [], 0
    for g0 in range(ng0):
        bnd = lo + ((g0 + 1) * span) // ng0
        v0 = v
        while v < len(slots) and slots[v] < bnd:
            v += 1
        if v > v0:
            out.append((v0, v))
            assert slots[v - 1] - slots[v0] + 1 <= GROUP_SLOTS
    return out


def split_track(prob, j):
    """(narrow blocks [(H, r)], remainder block (H, r)) of track j: rows over the full state, two-level basis."""
    r, H_x, H_f = oracle.feature_blocks(prob, j)
    a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
    slots = np.asarray(prob.obs_slot[a:b])
    narrow, cH, cr, cf = [], [], [], []
    for v0, v1 in groups_of(slots):
        rows = np.arange(2 * v0, 2 * v1)
        Q = householder_q(H_f[rows])
        Hq, rq, fq = Q.T @ H_x[rows], Q.T @ r[rows], Q.T @ H_f[rows]
        c = min(3, len(rows))
        assert np.abs(fq[c:]).max(initial=0.0) < 1e-12 * np.abs(H_f).max()
        cH.append(Hq[:c]); cr.append(rq[:c]); cf.append(fq[:c])
        if len(rows) > c:
            narrow.append((Hq[c:], rq[c:]))
    CH, Cr, Cf = np.vstack(cH), np.concatenate(cr), np.vstack(cf)
    from scipy.linalg import null_space
    A2 = null_space(Cf.T)                                   # level 2: the stacked carries, rank rule of MSCKF.py:555
    return narrow, (A2.T @ CH, A2.T @ Cr)


def update_with_split(prob):
    P, d, s2 = prob.P, prob.P.shape[0], prob.sigma ** 2
    Hs, rs, acc, n_narrow, n_rem = [], [], np.zeros(prob.F, np.uint8), 0, 0
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        slots = np.asarray(prob.obs_slot[a:b])
        if slots.max() - slots.min() + 1 > GROUP_SLOTS:
            narrow, rem = split_track(prob, j)
            for Hn, _ in narrow:                            # a narrow block touches its group's slots only
                used = np.nonzero(np.abs(Hn).sum(axis=0) > 0)[0]
                assert used.min() >= 15 and (used.max() - 15) // 6 - (used.min() - 15) // 6 + 1 <= GROUP_SLOTS
            H = np.vstack([h for h, _ in narrow] + [rem[0]])
            rr = np.concatenate([x for _, x in narrow] + [rem[1]])
            n_narrow += sum(h.shape[0] for h, _ in narrow); n_rem += rem[0].shape[0]
        else:
            rj, Hx, Hf = oracle.feature_blocks(prob, j)
            rr, H = oracle.project_on_nullspace(Hf, rj, Hx)
        ok, _, _ = oracle.gate(rr, H, P, prob.sigma)
        if ok:
            acc[j] = 1
            Hs.append(H); rs.append(rr)
    H, r = np.vstack(Hs), np.concatenate(rs)
    Q, R = np.linalg.qr(H)
    S = R @ P @ R.T + s2 * np.eye(R.shape[0])
    K = P @ R.T @ np.linalg.inv(S)
    IKT = np.eye(d) - K @ R
    Pn = IKT @ P @ IKT.T + s2 * K @ K.T
    return K @ (Q.T @ r), (Pn + Pn.T) / 2, acc, n_narrow, n_rem


@pytest.mark.parametrize("N,F,M,seed,kw", [
    (30, 40, 30, 1, {}),
    (30, 60, 30, 2, {"variable_tracks": True, "min_track": 2, "outlier_fraction": 0.1, "outlier_px": 400.0}),
    (31, 24, 31, 3, {}),
    (16, 30, 16, 4, {}),
    (12, 40, 12, 5, {"variable_tracks": True, "min_track": 2}),
])
def test_two_level_basis_gives_the_oracles_update(N, F, M, seed, kw):
    prob = synth.make_problem(N, F, M, seed=seed, **kw)
    ref = oracle.update(prob, dense_noise=False)
    dx, Pn, acc, n_narrow, n_rem = update_with_split(prob)
    assert np.array_equal(acc, ref["accepted"])
    assert rel_err(dx, ref["dx"]) < 1e-11 and rel_err(Pn, ref["P_new"]) < 1e-12
    assert n_rem > 0 and n_narrow > 4 * n_rem              # most rows of a long track are narrow


def test_two_level_basis_on_the_reference_fixture_with_a_gauge_prior():
    """edge_gauge_prior: 100 m^2 of common-mode position variance, exactly in the stack's null space."""
    prob, ref = load_golden("edge_gauge_prior")
    dx, Pn, acc, _, _ = update_with_split(prob)
    assert np.array_equal(acc, ref["accepted"])
    assert rel_err(dx, ref["dx"]) < 1e-9 and rel_err(Pn, ref["P_new"]) < 1e-11
