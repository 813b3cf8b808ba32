"""f4: the geometric consistency tests of the front end's matches on the device
(k_assoc, msckf_run_associate) against the reference's own run (fixture assoc_tests) and the oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_association_fixture():
    from msckf_amd.api import UpdateEngine
    prob, z = load_golden("assoc_tests")
    with UpdateEngine(max_clones=prob.N, max_features=prob.F, max_track=8) as e:
        e.load(prob)
        res, fail = e.associate(z["assoc_matched_uv"], z["assoc_R_cur"], z["assoc_t_cur"], prob.K,
                                float(z["assoc_thr"][0]), float(z["assoc_thr"][1]))
    assert np.array_equal((res == 0).astype(np.uint8), z["assoc_kept"])                  # the views the reference appended
    assert int((res == 1).sum()) == int(z["assoc_n_epipolar"])
    assert int((res == 2).sum()) == int(z["assoc_n_homography"])
    assert np.array_equal(res == 3, np.isnan(z["assoc_matched_uv"][:, 0]))


@pytest.mark.parametrize("N,F,M,seed", [(30, 5000, 10, 1), (12, 300, 12, 2)])
def test_association_against_oracle(N, F, M, seed):
    """Flags and failing view bit-equal to the oracle on random matches (thresholds placed between the scores)."""
    from msckf_amd import synth
    from msckf_amd.api import UpdateEngine
    from oracle import msckf_oracle as oracle
    rng = np.random.default_rng(seed)
    prob = synth.make_problem(N, F, M, seed=seed, variable_tracks=True, min_track=1)
    R_cur = synth.so3_exp(0.03 * rng.standard_normal(3)) @ prob.cam_R[-1]
    t_cur = prob.cam_t[N // 2] + np.where(rng.uniform() < 2, 2e-3, 0.0) * rng.standard_normal(3)   # homography against clone N/2
    muv = np.column_stack([rng.uniform(0, 640, F), rng.uniform(0, 480, F)])
    muv[rng.uniform(size=F) < 0.1] = np.nan
    seen = set()
    with UpdateEngine(max_clones=N, max_features=F, max_track=M) as e:
        e.load(prob)
        for thr_e, thr_h in ((1e-3, 150.0), (1e9, 0.5)):         # the second pair lets every match reach the 1 cm clone
            ref, rfail = oracle.associate(prob, muv, R_cur, t_cur, prob.K, thr_e, thr_h)
            res, fail = e.associate(muv, R_cur, t_cur, prob.K, thr_e, thr_h)
            assert np.array_equal(res, ref) and np.array_equal(fail, rfail)
            seen |= set(np.unique(res).tolist())
    assert seen == {0, 1, 2, 3}
