"""Long tracks (more than 10 clone slots) through the split of round 5: a two-level nullspace basis
makes 2 M_g - 3 rows of every view group an ordinary <= 10-slot track of the 60-column band pipeline and leaves
3 (groups - 1) remainder rows per track, which K6-K7 takes as they are (few) or as the root of a merge tree of their own
(many).  Everything here goes through the C-ABI and is compared with the oracle at 1e-8 (BASELINE.json).

reference: MSCKF.py:554-559 (any orthonormal basis of null(H_f^T) gives the same dx / P+), :594-614."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from msckf_amd import synth
from oracle import msckf_oracle as oracle

pytestmark = pytest.mark.gpu

TOL = 1e-8


def _check(eng, prob, ref=None, tol=TOL):
    ref = ref or oracle.update(prob, dense_noise=False)
    res = eng.update_problem(prob)
    assert res.status == ref["status"]
    assert np.array_equal(res.accepted, ref["accepted"])
    assert rel_err(res.dx, ref["dx"]) < tol, rel_err(res.dx, ref["dx"])
    assert rel_err(res.P_new, ref["P_new"]) < tol, rel_err(res.P_new, ref["P_new"])
    assert np.array_equal(res.P_new, res.P_new.T)
    return res, ref


@pytest.mark.parametrize("N", [11, 12, 20, 30, 31])
def test_first_call_of_a_fresh_engine(N):
    """the FIRST long-track call of a freshly created engine, per window size (round 4's hand-off of the
    information form failed exactly there).  25 engines per N, each compared with the oracle."""
    from msckf_amd.api import UpdateEngine
    prob = synth.make_problem(N, 48, N, seed=300 + N, variable_tracks=True, min_track=2)
    ref = oracle.update(prob, dense_noise=False)
    for _ in range(25):
        with UpdateEngine(max_clones=N, max_features=64, max_track=N) as e:
            _check(e, prob, ref)


@pytest.fixture(scope="module")
def eng():
    from msckf_amd.api import UpdateEngine
    e = UpdateEngine(max_clones=53, max_features=4096, max_track=31)
    yield e
    e.close()


def test_the_plans_the_split_chooses(eng):
    """few long tracks: remainder rows taken as they are; many: the same (up to 3840 rows); forced small limit: their own tree;
    a batch most of whose tracks span 11 - 15 slots and none more (BASELINE configs[4]): no split, 90-column tiles."""
    p = synth.few_long_tracks_problem(30, 400, 10, 10, seed=1)
    _check(eng, p)
    s = eng.debug_split()
    assert s["long_tracks"] == 10 and s["narrow_blocks"] == 30 and s["remainder_rows_cap"] == 90 and s["remainder_mode"] == 1
    assert s["band_plan"] == 1 and s["sweep_mode"] == 0 and s["entries"] == 400 + 30 + 10
    eng.set_rem_direct_rows(64)
    try:
        _check(eng, p)
        s = eng.debug_split()
        assert s["remainder_mode"] == 2 and s["remainder_tree_levels"] >= 1
    finally:
        eng.set_rem_direct_rows(-1)
    p15 = synth.make_problem(50, 300, 15, seed=2)
    _check(eng, p15)
    s = eng.debug_split()
    assert s["long_tracks"] == 0 and s["band_plan"] == 1 and s["sweep_mode"] == 2
    p10 = synth.make_problem(30, 300, 10, seed=3)
    _check(eng, p10)
    assert eng.debug_split()["long_tracks"] == 0 and eng.debug_split()["entries"] == 300


@pytest.mark.parametrize("N,F,M,seed,kw,direct_rows", [
    (30, 300, 30, 61, {"variable_tracks": True, "min_track": 2, "outlier_fraction": 0.1, "outlier_px": 400.0}, -1),
    (30, 300, 30, 61, {"variable_tracks": True, "min_track": 2, "outlier_fraction": 0.1, "outlier_px": 400.0}, 0),   # remainder tree
    (31, 64, 31, 62, {}, -1),
    (31, 64, 31, 62, {}, 0),
    (53, 200, 31, 63, {"variable_tracks": True, "min_track": 2}, -1),          # ring-buffered band root + dense second source
    (53, 200, 31, 63, {"variable_tracks": True, "min_track": 2}, 0),
    (40, 150, 25, 64, {"variable_tracks": True, "min_track": 11}, -1),         # every track long: no short track at all
    (16, 80, 16, 65, {}, 0),
    (11, 60, 11, 66, {"variable_tracks": True, "min_track": 2}, -1),           # the shortest long track: 11 slots -> groups of 6 + 5
    (20, 1, 20, 67, {}, -1),                                                   # a single long track: m < 6N
    (20, 1, 20, 67, {}, 0),
])
def test_long_tracks_against_oracle(eng, N, F, M, seed, kw, direct_rows):
    prob = synth.make_problem(N, F, M, seed=seed, **kw)
    eng.set_rem_direct_rows(direct_rows)
    try:
        res, ref = _check(eng, prob)
        s = eng.debug_split()
        assert s["long_tracks"] > 0 and s["remainder_mode"] == (1 if direct_rows < 0 else 2)
        gam, q = eng.debug_gate()
        np.testing.assert_allclose(gam, ref["gamma"], rtol=1e-8, atol=1e-12)     # (the oracle inverts S explicitly; 400 px outliers)
        # the compressed system (both sources of rows together): T^T T = H^T H, T^T r_n = H^T r
        T, rn = eng.debug_compressed()
        H, r = ref["H_X"][:, 15:], ref["r_o"]
        assert rel_err(T.T @ T, H.T @ H) < 1e-10 and rel_err(T.T @ rn, H.T @ r) < 1e-10
        assert res.stats["stacked_rows"] == H.shape[0]
        # resident sequence: the same batch three times back to back
        eng.load(prob)
        for _ in range(3):
            eng.run()
        r2 = eng.result()
        assert rel_err(r2.dx, ref["dx"]) < TOL and rel_err(r2.P_new, ref["P_new"]) < TOL
    finally:
        eng.set_rem_direct_rows(-1)


@pytest.mark.parametrize("cut_rows,levels", [(None, None), ("0", 4), ("200", 3), ("1000", 1)])
def test_the_remainder_tree_cut_and_whole(monkeypatch, cut_rows, levels):
    """Many remainder rows: their merge tree ends where a further level would remove fewer rows than K6-K7 takes in the launch's time,
    and K6-K7 takes the triangles it ends with as dense rows (default: a level must remove 600 rows); MSCKF_REM_CUT_ROWS=0: the
    whole tree (8 leaves, three merge levels) and its root block; 200: all but the last level; 1000: the leaves' triangles.  All against
    the oracle (reference MSCKF.py:594-614)."""
    from msckf_amd.api import UpdateEngine
    if cut_rows is None:
        monkeypatch.delenv("MSCKF_REM_CUT_ROWS", raising=False)
    else:
        monkeypatch.setenv("MSCKF_REM_CUT_ROWS", cut_rows)
    prob = synth.make_problem(30, 500, 30, seed=71, variable_tracks=True, min_track=2, outlier_fraction=0.05, outlier_px=300.0)
    with UpdateEngine(max_clones=30, max_features=512, max_track=30) as e:
        e.set_rem_direct_rows(0)
        res, ref = _check(e, prob)
        s = e.debug_split()
        assert s["remainder_mode"] == 2 and (s["remainder_tree_levels"] == levels if levels else 1 <= s["remainder_tree_levels"] <= 3), s
        T, rn = e.debug_compressed()                       # (one [T | r_n] of the band root and the second source's rows)
        H, r = ref["H_X"][:, 15:], ref["r_o"]
        assert rel_err(T.T @ T, H.T @ H) < 1e-10 and rel_err(T.T @ rn, H.T @ r) < 1e-10
        res2 = e.update_problem(prob)
        assert np.array_equal(res.dx, res2.dx) and np.array_equal(res.P_new, res2.P_new)


@pytest.mark.parametrize("case", ["edge_long_tracks", "edge_mixed_spans", "edge_few_long_among_short", "edge_long_tracks_B",
                                  "edge_gauge_prior", "edge_few_rows_long_tracks"])
@pytest.mark.parametrize("direct_rows", [-1, 0])
def test_reference_fixtures_on_both_remainder_paths(eng, case, direct_rows):
    """The reference's own outputs (tests/golden/gen_golden.py) for long tracks, with the remainder rows taken as they are
    and through their own tree; edge_gauge_prior: a 10 m common-mode prior, where a normal-equation shortcut would show."""
    prob, ref = load_golden(case)
    eng.set_rem_direct_rows(direct_rows)
    try:
        res = eng.update_problem(prob)
        assert res.status == int(ref["status"]) and np.array_equal(res.accepted, ref["accepted"])
        assert rel_err(res.dx, ref["dx"]) < TOL and rel_err(res.P_new, ref["P_new"]) < TOL
        assert eng.debug_split()["long_tracks"] > 0
    finally:
        eng.set_rem_direct_rows(-1)


def test_views_out_of_slot_order_are_not_split(eng):
    """A long track whose views do not come in slot order keeps one block (the groups of a split are view ranges): the
    batch falls back to one Householder plan for every track.  Same result."""
    prob = synth.make_problem(24, 60, 24, seed=70, variable_tracks=True, min_track=12)
    ref = oracle.update(prob, dense_noise=False)
    vp = prob.view_ptr
    uv, sl = prob.obs_uv.copy(), prob.obs_slot.copy()
    for j in range(0, prob.F, 2):                              # every second track: views reversed
        a, b = int(vp[j]), int(vp[j + 1])
        uv[a:b] = uv[a:b][::-1]
        sl[a:b] = sl[a:b][::-1]
    q = synth.UpdateProblem(**{**prob.__dict__, "obs_uv": uv, "obs_slot": sl})
    res = eng.update_problem(q)
    assert res.status == 0 and np.array_equal(res.accepted, ref["accepted"])
    assert rel_err(res.dx, ref["dx"]) < TOL and rel_err(res.P_new, ref["P_new"]) < TOL
    s = eng.debug_split()
    assert 0 < s["long_tracks"] < prob.F


def test_tracks_with_holes_and_one_view_groups(eng):
    """Long tracks whose views leave whole stretches of slots empty (a group without views is skipped) or a single view in
    a group (two carry rows, no narrow block)."""
    rng = np.random.default_rng(71)
    base = synth.make_problem(30, 120, 30, seed=71)
    keep_all = []
    vp = [0]
    for j in range(base.F):
        a = int(base.view_ptr[j])
        pat = j % 4
        if pat == 0:
            keep = [0, 1, 2, 15, 29]                           # one-view groups
        elif pat == 1:
            keep = [0, 1, 2, 3] + list(range(22, 30))          # an empty middle stretch
        elif pat == 2:
            keep = sorted(rng.choice(30, size=int(rng.integers(4, 20)), replace=False).tolist())
        else:
            keep = list(range(30))
        keep_all += [a + k for k in keep]
        vp.append(vp[-1] + len(keep))
    idx = np.array(keep_all)
    q = synth.UpdateProblem(**{**base.__dict__, "view_ptr": np.array(vp, dtype=np.int32), "obs_uv": base.obs_uv[idx],
                               "obs_slot": base.obs_slot[idx]})
    _check(eng, q)
    assert eng.debug_split()["long_tracks"] >= 60


def test_select_then_update_with_long_tracks(eng):
    """get_valid_features -> update (MSCKF.py:450-456) on a batch with long tracks: the blocks of a track k_select left out
    carry no rows, the others go through the split."""
    prob = synth.make_problem(30, 200, 30, seed=72, variable_tracks=True, min_track=2)
    tracks = synth.make_tracks(prob, seed=72, lost_fraction=0.6)
    sp = synth.SelectParams(min_parallax_deg=4.0)
    sel = oracle.select_features(prob, tracks, sp)
    valid = np.nonzero(sel["flags"] & 1)[0]
    assert 20 < len(valid) < prob.F
    sub = synth.UpdateProblem(**{**prob.__dict__, "idp_m": sel["idp_m"], "idp_rho": sel["idp_rho"]}).take(valid)
    ref = oracle.update(sub, dense_noise=False)
    eng.load(prob)
    eng.set_tracks(tracks)
    eng.run_select(sp, prob.K)
    eng.run()
    res = eng.result()
    assert res.status == ref["status"]
    assert np.array_equal(res.accepted[valid], ref["accepted"]) and not res.accepted[np.setdiff1d(np.arange(prob.F), valid)].any()
    assert rel_err(res.dx, ref["dx"]) < TOL and rel_err(res.P_new, ref["P_new"]) < TOL
    eng.replan()                                               # the plan over the valid tracks only
    eng.run()
    res2 = eng.result()
    assert rel_err(res2.dx, ref["dx"]) < TOL and rel_err(res2.P_new, ref["P_new"]) < TOL


def test_exported_block_of_a_batch_with_long_tracks(eng):
    """msckf_run_compress: the block leaves the context, so every block of the split goes into ONE plan (merge tree)."""
    prob = synth.make_problem(30, 150, 30, seed=73, variable_tracks=True, min_track=2)
    ref = oracle.update(prob, dense_noise=False)
    eng.load(prob)
    eng.run_compress()
    blk, n_acc = eng.export_block()
    assert n_acc == int(ref["accepted"].sum())
    T, rn = blk[:, :-1], blk[:, -1]
    H, r = ref["H_X"][:, 15:], ref["r_o"]
    assert rel_err(T.T @ T, H.T @ H) < 1e-10 and rel_err(T.T @ rn, H.T @ r) < 1e-10
    assert eng.debug_split()["remainder_mode"] == 0
    eng.run()                                                   # ... and the update on that plan
    res = eng.result()
    assert rel_err(res.dx, ref["dx"]) < TOL and rel_err(res.P_new, ref["P_new"]) < TOL


def test_bitwise_repeatability_with_long_tracks(eng):
    """Rotating batches through the one-shot call: every result bit for bit that of the batch's first call AND within
    1e-8 of the oracle (round 4 compared repeats with the first result only)."""
    probs = [synth.make_problem(N, F, M, seed=80 + i, variable_tracks=True, min_track=2)
             for i, (N, F, M) in enumerate([(30, 200, 30), (31, 90, 31), (20, 150, 20), (30, 400, 12), (12, 60, 12), (30, 64, 30)])]
    refs = [oracle.update(p, dense_noise=False) for p in probs]
    first = [None] * len(probs)
    for it in range(120):
        i = (it * 5 + it // 7) % len(probs)
        res = eng.update_problem(probs[i])
        assert res.status == 0
        if first[i] is None:
            first[i] = (res.dx.copy(), res.P_new.copy())
            assert rel_err(res.dx, refs[i]["dx"]) < TOL and rel_err(res.P_new, refs[i]["P_new"]) < TOL
        else:
            assert np.array_equal(res.dx, first[i][0]) and np.array_equal(res.P_new, first[i][1]), (it, i)


def test_stress_ragged_long_tracks_against_oracle():
    """tools/stress_split.py, 40 rounds over 24 ragged long-track batches on ONE engine (960 one-shot calls): every result
    within 1e-8 of the oracle and bit for bit the batch's first.  (Found, at 1 call in 200: round 4's second source of rows
    laid its partial tiles out for the band's strip count -- k_gstream.h.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_split.py"), "40", "24"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "960 calls, 0 bad" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
