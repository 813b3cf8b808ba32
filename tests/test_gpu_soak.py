"""Randomised parity soak (promoted from tools/ in round 3): seeded random (N, F, M) shapes -- ragged tracks,
outliers, every K5 plan (60-column sweeps, ring, 90-column tiles, merge tree), both K6 forms -- through the
one-shot drop-in call against the oracle.  Tolerance 1e-8 relative on dx and P+ (BASELINE.json), mask bit-equal."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-8


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        N = int(rng.integers(2, 54))
        M = int(rng.integers(2, min(N, 20) + 1))
        F = int(rng.integers(1, 400))
        kw = {}
        if rng.random() < 0.4:
            kw["variable_tracks"] = True
        if rng.random() < 0.3:
            kw.update(outlier_fraction=0.1, outlier_px=300.0)
        out.append((N, F, M, int(rng.integers(0, 10 ** 6)), kw))
    return out


@pytest.mark.parametrize("seed", [7, 8])
def test_random_shapes_against_oracle(seed):
    from msckf_amd import synth
    from msckf_amd.api import UpdateEngine
    from oracle import msckf_oracle as oracle
    worst = (0.0, 0.0)
    with UpdateEngine(max_clones=53, max_features=400, max_track=20) as e:
        for (N, F, M, sd, kw) in _cases(seed, 30):
            prob = synth.make_problem(N, F, M, seed=sd, **kw)
            ref = oracle.update(prob, dense_noise=False)
            r = e.update_problem(prob)
            tag = dict(N=N, F=F, M=M, seed=sd, **kw)
            assert r.status == ref["status"], tag
            assert np.array_equal(r.accepted, ref["accepted"]), tag
            if ref["status"] == 0:
                edx, eP = rel_err(r.dx, ref["dx"]), rel_err(r.P_new, ref["P_new"])
                assert edx < TOL and eP < TOL, (tag, edx, eP)
                worst = (max(worst[0], edx), max(worst[1], eP))
    print("soak worst dx %.2e P %.2e" % worst)


def _more_cases(seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(40):
        N = int(rng.integers(2, 54)); M = int(rng.integers(2, min(N, 20) + 1)); F = int(rng.integers(1, 600))
        kw = {}
        if rng.random() < 0.5:
            kw["variable_tracks"] = True
        if rng.random() < 0.3:
            kw.update(outlier_fraction=0.1, outlier_px=300.0)
        out.append((N, F, M, int(rng.integers(0, 10 ** 6)), kw))
    return out


@pytest.mark.parametrize("seed", list(range(20, 28)))
def test_random_shapes_320(seed):
    """The 320-shape soak of round 3's tools/soak_more.py as a test: it runs on every GPU pass, i.e. after every change of
    the kernels' arithmetic (round 3's last three k_feature changes landed behind its last manual run)."""
    from msckf_amd import synth
    from msckf_amd.api import UpdateEngine
    from oracle import msckf_oracle as oracle
    with UpdateEngine(max_clones=53, max_features=600, max_track=20) as e:
        for (N, F, M, sd, kw) in _more_cases(seed):
            prob = synth.make_problem(N, F, M, seed=sd, **kw)
            ref = oracle.update(prob, dense_noise=False)
            r = e.update_problem(prob)
            tag = dict(N=N, F=F, M=M, seed=sd, **kw)
            assert r.status == ref["status"], tag
            assert np.array_equal(r.accepted, ref["accepted"]), tag
            if ref["status"] == 0:
                edx, eP = rel_err(r.dx, ref["dx"]), rel_err(r.P_new, ref["P_new"])
                assert edx < TOL and eP < TOL, (tag, edx, eP)
