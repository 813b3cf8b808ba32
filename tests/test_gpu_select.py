"""Parity of f1: the device `get_valid_features` (k_select) and the fused
select -> update pass, called through the C-ABI, against the fixtures captured from the
reference's own `get_valid_features` + `update`, and against the oracle on seeded problems.

Tolerances: flags bit-exact; inverse-depth points and triangulated points within
200 eps cond(X) relative (floor 1e-12) (the 3x3 normal matrix X of `intersection_of_lines` is solved by
pinv in both, forward error ~ eps cond) and never looser than 1e-8; dx and P+ of the chained
update within 1e-8 relative (BASELINE.json north_star)."""
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import load_golden_select, rel_err, select_cases

pytestmark = pytest.mark.gpu

TOL = 1e-8
EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def eng():
    from msckf_amd.api import UpdateEngine
    e = UpdateEngine(max_clones=31, max_features=12000, max_track=31)
    yield e
    e.close()


def check_selection(sel, exp_flags, exp_m, exp_rho, exp_world, cond):
    assert np.array_equal(sel.flags, exp_flags)
    tol = np.minimum(np.maximum(200 * EPS * cond, 1e-12), TOL)
    ref_mask = (exp_flags & 4) > 0
    assert np.all(np.abs(sel.idp_rho - exp_rho) <= tol * np.abs(exp_rho))
    assert np.all(np.abs(sel.idp_m - exp_m).max(axis=1) <= tol)
    scale = np.maximum(np.linalg.norm(exp_world[ref_mask], axis=1), 1.0)
    assert np.all(np.linalg.norm(sel.world[ref_mask] - exp_world[ref_mask], axis=1) <= tol[ref_mask] * scale)
    # features that were not refreshed keep their inverse-depth point bit for bit
    keep = ~ref_mask
    assert np.array_equal(sel.idp_rho[keep], exp_rho[keep]) and np.array_equal(sel.idp_m[keep], exp_m[keep])


@pytest.mark.parametrize("case", select_cases())
def test_select_and_chained_update_match_reference(eng, case):
    from oracle import msckf_oracle as oracle
    prob, tracks, params, ref = load_golden_select(case)
    cond = oracle.select_features(prob, tracks, params)["cond"]
    eng.load(prob)
    eng.set_tracks(tracks)
    eng.run_select(params, prob.K)
    eng.run()
    sel = eng.selection()
    check_selection(sel, ref["sel_flags"], ref["sel_idp_m"], ref["sel_idp_rho"], ref["sel_world"], cond)
    res = eng.result()
    assert res.status == int(ref["status"])
    assert res.n_rejected == int(ref["n_rejected"])
    assert not res.accepted[~sel.valid].any()
    if res.status == 0:
        assert rel_err(res.dx, ref["dx"]) < TOL
        assert rel_err(res.P_new, ref["P_new"]) < TOL
    else:
        assert np.array_equal(res.P_new, prob.P) and not res.dx.any()


@pytest.mark.parametrize("N,F,M,seed,pk,tk,sp", [
    (30, 2000, 10, 41, {}, {"lost_fraction": 0.4}, {"min_parallax_deg": 8.0}),                 # headline size
    (30, 10000, 10, 42, {"outlier_fraction": 0.05}, {"lost_fraction": 0.2}, {"min_parallax_deg": 12.0}),
    (16, 600, 16, 43, {"variable_tracks": True, "min_track": 1}, {"lost_fraction": 0.7, "flip_fraction": 0.2},
     {"min_frames_tracked": 3, "min_parallax_deg": 3.0, "width": 400, "height": 300}),
    (31, 200, 31, 44, {}, {"lost_fraction": 1.0}, {"use_parallax": False}),                   # 31 lines per feature
])
def test_select_against_oracle(eng, N, F, M, seed, pk, tk, sp):
    from msckf_amd import synth
    from oracle import msckf_oracle as oracle
    prob = synth.make_problem(N, F, M, seed=seed, **pk)
    tracks = synth.make_tracks(prob, seed, **tk)
    params = synth.SelectParams(**sp)
    exp = oracle.select_features(prob, tracks, params)
    eng.load(prob)
    eng.set_tracks(tracks)
    eng.run_select(params, prob.K)
    eng.run()
    sel = eng.selection()
    check_selection(sel, exp["flags"], exp["idp_m"], exp["idp_rho"], exp["world"], exp["cond"])
    res = eng.result()
    assert 0.0 < eng.time_select(5) < 1000.0           # re-launching is idempotent ...
    again = eng.selection()
    assert np.array_equal(again.flags, sel.flags) and np.array_equal(again.idp_rho, sel.idp_rho)
    valid = np.nonzero(exp["flags"] & 1)[0]
    chained = prob.take(valid)
    chained.idp_m, chained.idp_rho = exp["idp_m"][valid], exp["idp_rho"][valid]
    out = oracle.update(chained)
    assert res.status == out["status"] and res.n_rejected == out["n_rejected"]
    assert np.array_equal(res.accepted[valid], out["accepted"])
    assert rel_err(res.dx, out["dx"]) < TOL and rel_err(res.P_new, out["P_new"]) < TOL


def test_fused_pass_equals_two_calls(eng):
    """select -> run on the full batch == update_problem on the valid subset with refreshed points."""
    prob, tracks, params, _ = load_golden_select("sel_parallax5")
    sel = eng.select_problem(prob, tracks, params)
    eng.run()
    fused = eng.result()
    valid = np.nonzero(sel.valid)[0]
    two = prob.take(valid)
    two.idp_m, two.idp_rho = sel.idp_m[valid], sel.idp_rho[valid]
    sep = eng.update_problem(two)
    assert np.array_equal(fused.accepted[valid], sep.accepted)
    assert rel_err(fused.dx, sep.dx) < 1e-11 and rel_err(fused.P_new, sep.P_new) < 1e-12


@pytest.mark.parametrize("lost_fraction", [0.5, 0.03, 0.0])
def test_replan_over_valid_features_only(eng, lost_fraction):
    """msckf_replan shrinks the QR tree to the valid features; same update, fewer levels."""
    from msckf_amd import synth
    prob = synth.make_problem(30, 1500, 10, seed=51)
    tracks = synth.make_tracks(prob, 51, lost_fraction=lost_fraction)
    params = synth.SelectParams(use_parallax=False)
    sel = eng.select_problem(prob, tracks, params)
    eng.run()
    full = eng.result()
    eng.replan()
    eng.run()
    again = eng.result()
    assert again.status == full.status and np.array_equal(again.accepted, full.accepted)
    assert again.stats["n_leaves"] <= full.stats["n_leaves"]
    if full.status == 0:
        assert rel_err(again.dx, full.dx) < 1e-10 and rel_err(again.P_new, full.P_new) < 1e-11
        if lost_fraction < 0.1:
            assert again.stats["n_levels"] < full.stats["n_levels"]
    else:
        assert not sel.valid.any() and again.stats["n_leaves"] == 0
    # sharded export after a replan (also the empty tree) stays consistent
    eng.run_compress()
    blk, n = eng.export_block()
    assert n == int(full.accepted.sum()) and np.allclose(np.tril(blk[:, :-1], -1), 0.0)


def test_clear_selection_and_call_order(eng):
    from msckf_amd import _ffi
    prob, tracks, params, _ = load_golden_select("sel_default")
    eng.load(prob)
    with pytest.raises(_ffi.EngineError) as e:
        eng.run_select(params, prob.K)                  # no tracks uploaded for this batch
    assert e.value.code == _ffi.ERR_STATE
    eng.set_tracks(tracks)
    eng.run_select(params, prob.K)
    eng.run()
    n_sel = int(eng.result().stats["n_features"])
    assert n_sel == int(eng.selection().valid.sum()) < prob.F
    eng.clear_selection()
    eng.run()
    assert int(eng.result().stats["n_features"]) == prob.F
    eng.set_features(prob)                              # a new batch drops the tracks and the selection
    with pytest.raises(_ffi.EngineError):
        eng.selection()


def test_process_features_on_reference_shaped_objects(eng):
    """`UpdateEngine.process_features(filt)` mutates reference-shaped objects the way
    MSCKF.process_features does (MSCKF.py:450-456): refreshed inverse-depth points, covariance,
    counter, and lost features handed to the filter's own remove_features."""
    prob, tracks, params, ref = load_golden_select("sel_variable_tracks")
    keys = [5 * (i + 2) for i in range(prob.N)]
    cams = OrderedDict()
    for i, k in enumerate(keys):
        pose = SimpleNamespace(R=prob.cam_R[i].copy(), t=prob.cam_t[i].copy())
        cams[k] = SimpleNamespace(T_W_Ci=pose, T_W_Ci_null=pose, width=params.width, height=params.height)
    imu = SimpleNamespace(W_gravity=prob.gravity.copy(), T_W_Ii=SimpleNamespace(R=np.eye(3), t=np.zeros(3)),
                          v_W_Ii=np.zeros(3), gyroscope_bias=np.zeros(3), accelerometer_bias=np.zeros(3))
    feats = OrderedDict()
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        feats[300 + j] = SimpleNamespace(
            keypoints=[prob.obs_uv[i].copy() for i in range(a, b)],
            camera_indices=[keys[int(prob.obs_slot[i])] for i in range(a, b)],
            lines=[SimpleNamespace(base=tracks.line_base[i], direction=tracks.line_dir[i], confidence=tracks.line_conf[i])
                   for i in range(a, b)],
            lost_for_n_frames=int(tracks.lost_for[j]), tracked_for_n_frames=int(tracks.tracked_for[j]),
            inverse_depth_point=SimpleNamespace(base=prob.idp_base[j].copy(), m=prob.idp_m[j].copy(),
                                                rho=float(prob.idp_rho[j])))
    removed = {}
    filt = SimpleNamespace(
        state=SimpleNamespace(cameras=cams, covariance=prob.P.copy(), imu=imu), K=prob.K, sigma_image=prob.sigma,
        features=feats, number_of_residuals_discarded_for_gasting_test=0, estimated_world_points=[],
        min_number_of_frames_to_be_lost=params.min_frames_lost, min_number_of_frames_to_be_tracked=max(params.min_frames_tracked, 2),
        use_parallax=params.use_parallax, min_parallax=params.min_parallax_deg, remove_features=removed.update)
    assert eng.process_features(filt) == int(ref["status"])
    assert rel_err(filt.state.covariance, ref["P_new"]) < TOL
    assert filt.number_of_residuals_discarded_for_gasting_test == int(ref["n_rejected"])
    assert sorted(removed) == [300 + j for j in np.nonzero(ref["sel_flags"] & 2)[0]]
    assert len(filt.estimated_world_points) == int(((ref["sel_flags"] & 4) > 0).sum())
    rho = np.array([ft.inverse_depth_point.rho for ft in feats.values()])
    np.testing.assert_allclose(rho, ref["sel_idp_rho"], rtol=1e-8)


def _filter_from_fixture(prob, tracks, params, with_lcm=False):
    """A reference-shaped filter object (the attributes the pruning methods read) built from a sel_* fixture."""
    keys = [10 * (i + 1) for i in range(prob.N)]
    cams = OrderedDict()
    for i, k in enumerate(keys):
        pose = SimpleNamespace(R=prob.cam_R[i].copy(), t=prob.cam_t[i].copy())
        cams[k] = SimpleNamespace(T_W_Ci=pose, T_W_Ci_null=pose, width=params.width, height=params.height)
    imu = SimpleNamespace(W_gravity=prob.gravity.copy(), T_W_Ii=SimpleNamespace(R=np.eye(3), t=np.zeros(3)),
                          v_W_Ii=np.zeros(3), gyroscope_bias=np.zeros(3), accelerometer_bias=np.zeros(3))
    feats = OrderedDict()
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        feats[100 + j] = SimpleNamespace(
            keypoints=[prob.obs_uv[i].copy() for i in range(a, b)],
            camera_indices=[keys[int(prob.obs_slot[i])] for i in range(a, b)],
            descriptors=[None] * (b - a), scores=[0.0] * (b - a),
            lines=[SimpleNamespace(base=tracks.line_base[i], direction=tracks.line_dir[i], confidence=tracks.line_conf[i])
                   for i in range(a, b)],
            lost_for_n_frames=int(tracks.lost_for[j]), tracked_for_n_frames=int(tracks.tracked_for[j]),
            inverse_depth_point=SimpleNamespace(base=prob.idp_base[j].copy(), m=prob.idp_m[j].copy(),
                                                rho=float(prob.idp_rho[j])))
    filt = SimpleNamespace(
        state=SimpleNamespace(cameras=cams, covariance=prob.P.copy(), imu=imu), K=prob.K, sigma_image=prob.sigma,
        features=feats, number_of_residuals_discarded_for_gasting_test=0, estimated_world_points=[],
        min_number_of_frames_to_be_lost=params.min_frames_lost, min_number_of_frames_to_be_tracked=max(params.min_frames_tracked, 2),
        use_parallax=params.use_parallax, min_parallax=params.min_parallax_deg, last_camera_measurement=None)
    if with_lcm:
        ids = np.array(sorted(feats.keys()), dtype=np.int64)
        filt.last_camera_measurement = SimpleNamespace(
            descriptors=np.arange(len(ids) * 4, dtype=np.float64).reshape(len(ids), 4), features_indices=ids.copy())
    return filt, keys, feats


def _check_pruned(eng, filt, keys, feats, prob, ref, status):
    assert status == int(ref["prune_status"])
    left = list(filt.state.cameras.keys())
    assert [i for i, k in enumerate(keys) if k not in left] == list(ref["prune_removed_slots"])
    assert filt.state.covariance.shape == ref["prune_P_after"].shape
    assert rel_err(filt.state.covariance, ref["prune_P_after"]) < TOL
    for i, k in enumerate(left):
        np.testing.assert_allclose(filt.state.cameras[k].T_W_Ci.R, ref["prune_post_cam_R"][i], atol=1e-9)
        np.testing.assert_allclose(filt.state.cameras[k].T_W_Ci.t, ref["prune_post_cam_t"][i], atol=1e-9)
    assert filt.number_of_residuals_discarded_for_gasting_test == int(ref["prune_n_rejected"])
    assert len(filt.features) == int(ref["prune_features_left"])
    views = np.array([len(feats[100 + j].camera_indices) if (100 + j) in filt.features else 0 for j in range(prob.F)])
    assert np.array_equal(views, ref["prune_views_left"])
    assert eng.n_clones == prob.N - len(ref["prune_removed_slots"])     # the engine's resident state lost the clones as well


def test_prune_poorest_camera_states_composed_on_the_device(eng):
    """`UpdateEngine.prune_poorest_camera_states(filt)` against the reference's own run of
    `MSCKF.prune_poorest_camera_states` (MSCKF.py:710-737; fixture sel_prune_poorest): same clones removed, same
    covariance after update + removal, same poses of the remaining clones, same feature bookkeeping."""
    prob, tracks, params, ref = load_golden_select("sel_prune_poorest")
    filt, keys, feats = _filter_from_fixture(prob, tracks, params)
    status = eng.prune_poorest_camera_states(filt)
    _check_pruned(eng, filt, keys, feats, prob, ref, status)


def test_prune_camera_states_composed_on_the_device(eng):
    """`UpdateEngine.prune_camera_states(filt)` against the reference's own run of `MSCKF.prune_camera_states`
    (MSCKF.py:663-680; fixture sel_prune_states: every third clone of a 12-clone window): clones, covariance, poses,
    counters, and the `last_camera_measurement` entries of the features that lost all their views (:771-777)."""
    prob, tracks, params, ref = load_golden_select("sel_prune_states")
    filt, keys, feats = _filter_from_fixture(prob, tracks, params, with_lcm=True)
    filt.max_number_of_camera_states = int(ref["prune_max_states"])
    filt.camera_states_to_delete = int(ref["prune_states_to_delete"])
    status = eng.prune_camera_states(filt)
    _check_pruned(eng, filt, keys, feats, prob, ref, status)
    assert int(ref["prune_features_left"]) < prob.F                        # the fixture does exercise :771-777
    assert np.array_equal(filt.last_camera_measurement.features_indices, ref["prune_lcm_indices_left"])
    assert np.array_equal(filt.last_camera_measurement.descriptors, ref["prune_lcm_descriptors_left"])
