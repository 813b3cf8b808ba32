"""Parity of f2 / f3: the covariance stays in HBM across a multi-frame run --
propagate (process_imu), augment (state_augmentation), update + commit, remove clones
(remove_cameras) -- and must stay on the covariance the REFERENCE produced at every step of
the same run (fixtures `seq_*.npz`, captured by tests/golden/gen_golden.py), and on the oracle
for single steps at full window size.  Tolerance 1e-8 relative (BASELINE.json); observed ~1e-13."""
import numpy as np
import pytest

from conftest import load_sequence, rel_err, sequence_cases

pytestmark = pytest.mark.gpu

TOL = 1e-8


@pytest.fixture(scope="module")
def eng():
    from msckf_amd.api import UpdateEngine
    e = UpdateEngine(max_clones=32, max_features=4096, max_track=16)
    yield e
    e.close()


@pytest.mark.parametrize("case", sequence_cases())
def test_resident_run_tracks_the_reference(eng, case):
    from msckf_amd import propagation, synth
    head, ops = load_sequence(case)
    eng.set_prior(head["P0"], head["gravity"], head["K"], float(head["sigma"]))
    worst = 0.0
    for op in ops:
        if op["kind"] == 0:
            Phi, Q = propagation.imu_transition(op["R"], op["t"], op["v"], op["R0"], op["t0"], op["v0"], op["gyro"],
                                                op["acc"], float(op["dt"]), head["gravity"], head["Qc"], op["w_planet"])
            eng.propagate(Phi, Q)
        elif op["kind"] == 1:
            J, cR, ct = propagation.augmentation(op["imu_R"], op["imu_t"], (head["T_W_I_R"], head["T_W_I_t"]),
                                                 (head["T_W_C_R"], head["T_W_C_t"]))
            np.testing.assert_allclose(cR, op["cam_R"], atol=1e-13)
            np.testing.assert_allclose(ct, op["cam_t"], atol=1e-13)
            eng.augment(J, cR, ct)
        elif op["kind"] == 2:
            N = op["cam_R"].shape[0]
            assert eng.n_clones == N
            feats = synth.UpdateProblem(P=np.zeros((15 + 6 * N,) * 2), cam_R=op["cam_R"], cam_t=op["cam_t"],
                                        cam_R0=op["cam_R"], cam_t0=op["cam_t"], gravity=head["gravity"], K=head["K"],
                                        sigma=float(head["sigma"]), view_ptr=op["view_ptr"], obs_uv=op["obs_uv"],
                                        obs_slot=op["obs_slot"], idp_base=op["idp_base"], idp_m=op["idp_m"],
                                        idp_rho=op["idp_rho"])
            eng.set_features(feats)                      # only the batch travels; P and poses are resident
            eng.run()
            res = eng.result()
            assert res.status == int(op["status"])
            assert rel_err(res.dx, op["dx"]) < TOL
            assert eng.commit_covariance() == res.status
            eng.set_poses(op["post_cam_R"], op["post_cam_t"])         # poses after the host's injection
        else:
            eng.remove_clones(op["slots"])
        P = eng.covariance()
        assert P.shape == op["P_after"].shape
        err = rel_err(P, op["P_after"])
        worst = max(worst, err)
        assert err < TOL, (op["kind"], err)
        assert np.array_equal(P, P.T)
    assert worst < 1e-10


def random_state(N, seed):
    from msckf_amd import synth
    rng = np.random.default_rng(seed)
    prob = synth.make_problem(N, 10, min(N, 4), seed=seed)
    return prob, rng


def test_propagate_against_oracle_full_window(eng):
    from oracle import msckf_oracle as oracle
    prob, rng = random_state(30, 71)
    eng.set_prior(prob.P, prob.gravity, prob.K, prob.sigma, prob.cam_R, prob.cam_t)
    P = prob.P
    for k in range(5):
        Phi = np.eye(15) + 0.01 * rng.standard_normal((15, 15))
        A = rng.standard_normal((15, 15))
        Q = 1e-6 * A @ A.T
        eng.propagate(Phi, Q)
        P = oracle.propagate_covariance(P, Phi, Q)
    got = eng.covariance()
    assert rel_err(got, P) < 1e-13 and np.array_equal(got, got.T)


def test_propagate_symmetrises_like_the_reference(eng):
    """MSCKF.py:244 symmetrises the WHOLE matrix, also an asymmetric clone block."""
    from oracle import msckf_oracle as oracle
    prob, rng = random_state(6, 72)
    P = prob.P + 1e-6 * rng.standard_normal(prob.P.shape)
    eng.set_prior(P, prob.gravity, prob.K, prob.sigma, prob.cam_R, prob.cam_t)
    Phi, Q = np.eye(15) + 0.01 * rng.standard_normal((15, 15)), 1e-6 * np.eye(15)
    eng.propagate(Phi, Q)
    assert rel_err(eng.covariance(), oracle.propagate_covariance(P, Phi, Q)) < 1e-13


def test_augment_and_remove_against_oracle(eng):
    from oracle import msckf_oracle as oracle
    prob, rng = random_state(29, 73)
    eng.set_prior(prob.P, prob.gravity, prob.K, prob.sigma, prob.cam_R, prob.cam_t)
    J = np.zeros((6, 15))
    J[:3, :3] = rng.standard_normal((3, 3)); J[3:, :3] = rng.standard_normal((3, 3)); J[3:, 12:] = np.eye(3)
    eng.augment(J, np.eye(3), np.ones(3))
    P = oracle.augment_covariance(prob.P, J)
    assert eng.n_clones == 30 and rel_err(eng.covariance(), P) < 1e-14
    eng.remove_clones([28, 0, 7])
    P = oracle.remove_clones_covariance(P, [28, 0, 7])
    got = eng.covariance()
    assert eng.n_clones == 27 and np.array_equal(got, P)          # a pure gather: bit-exact


def test_update_after_window_change_uses_the_new_layout(eng):
    """augment -> remove -> update on the resident state == one-shot update on the same arrays."""
    from msckf_amd import synth
    from oracle import msckf_oracle as oracle
    prob, rng = random_state(9, 74)
    eng.set_prior(prob.P, prob.gravity, prob.K, prob.sigma, prob.cam_R, prob.cam_t)
    J = np.zeros((6, 15)); J[:3, :3] = np.eye(3); J[3:, 12:] = np.eye(3)
    new_R, new_t = prob.cam_R[-1], prob.cam_t[-1] + np.array([0.15, 0.0, 0.0])
    eng.augment(J, new_R, new_t)
    eng.remove_clones([2])
    P = oracle.remove_clones_covariance(oracle.augment_covariance(prob.P, J), [2])
    keep = [i for i in range(10) if i != 2]
    cam_R = np.concatenate([prob.cam_R, new_R[None]])[keep]
    cam_t = np.concatenate([prob.cam_t, new_t[None]])[keep]
    batch = synth.make_problem(9, 120, 5, seed=75, P=P, poses=(cam_R, cam_t))
    eng.set_features(batch)
    eng.run()
    res = eng.result()
    exp = oracle.update(batch)
    assert res.status == 0 and rel_err(res.dx, exp["dx"]) < TOL and rel_err(res.P_new, exp["P_new"]) < TOL


def test_window_errors(eng):
    from msckf_amd import _ffi
    prob, rng = random_state(32, 76)
    eng.set_prior(prob.P, prob.gravity, prob.K, prob.sigma, prob.cam_R, prob.cam_t)
    with pytest.raises(_ffi.EngineError) as e:
        eng.augment(np.zeros((6, 15)), np.eye(3), np.zeros(3))       # window full (max_clones = 32)
    assert e.value.code == _ffi.ERR_ARG
    for bad in ([32], [-1], [3, 3]):
        with pytest.raises(_ffi.EngineError) as e:
            eng.remove_clones(bad)
        assert e.value.code == _ffi.ERR_ARG
    eng.load(prob)
    eng.remove_clones([5])
    with pytest.raises(_ffi.EngineError) as e:
        eng.run()                                                   # the batch was planned for 32 clones
    assert e.value.code == _ffi.ERR_STATE
