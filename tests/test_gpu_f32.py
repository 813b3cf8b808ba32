"""BASELINE.json configs[4] mode: fp32 storage of the stacked system + Joseph covariance update on the f32
matrix cores (msckf_config.dtype = MSCKF_DTYPE_F32), against the fp64 oracle.

Tolerance of this mode: 1e-4 relative on dx, 1e-5 relative on P+ -- fp32 cannot meet the
1e-8 of the fp64 path.  Observed: ~2e-7 / ~1e-7.  The gate runs in fp64 before anything
is rounded, so the accepted mask is the reference's."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL_DX, TOL_P = 1e-4, 1e-5


@pytest.fixture(scope="module")
def eng32():
    from msckf_amd.api import UpdateEngine
    e = UpdateEngine(max_clones=53, max_features=20000, max_track=31, dtype="f32")
    yield e
    e.close()


@pytest.mark.parametrize("case", golden_cases())
def test_golden_f32(eng32, case):
    prob, ref = load_golden(case)
    res = eng32.update_problem(prob)
    assert res.status == int(ref["status"])
    assert np.array_equal(res.accepted, ref["accepted"])
    if res.status == 0:
        tol_dx, tol_p = TOL_DX, TOL_P
        if case == "edge_gauge_prior":
            # a 10 m common-mode position prior: the stack's exact null space (global translation + yaw) survives fp64
            # Householder rows, not their rounding to fp32 -- H u = 6e-8 |H| there, against a prior variance of 100 m^2.
            # NumPy with the oracle's stack rounded to fp32 gives 2.7e-2 on dx: this mode is for priors without
            # metre-level gauge variance; the fp64 engine meets 1e-8 on this fixture (test_golden)
            tol_dx, tol_p = 0.2, 1e-2
        assert rel_err(res.dx, ref["dx"]) < tol_dx
        assert rel_err(res.P_new, ref["P_new"]) < tol_p
        assert np.array_equal(res.P_new, res.P_new.T)
    else:
        assert np.array_equal(res.P_new, prob.P) and not res.dx.any()


@pytest.mark.parametrize("N,F,M,seed,kw", [
    (20, 500, 8, 71, {}),
    (30, 2000, 10, 72, {}),
    (30, 600, 15, 73, {"variable_tracks": True}),       # 90-column tiles, ragged
    (50, 1000, 15, 74, {"outlier_fraction": 0.05, "outlier_px": 500.0}),   # two-block K6 + ring
    (31, 64, 31, 75, {}),                               # merge tree
])
def test_f32_against_oracle(eng32, N, F, M, seed, kw):
    from msckf_amd import synth
    from oracle import msckf_oracle as oracle
    prob = synth.make_problem(N, F, M, seed=seed, **kw)
    ref = oracle.update(prob, dense_noise=False)
    res = eng32.update_problem(prob)
    assert res.status == ref["status"] == 0
    assert np.array_equal(res.accepted, ref["accepted"])
    assert rel_err(res.dx, ref["dx"]) < TOL_DX and rel_err(res.P_new, ref["P_new"]) < TOL_P
    assert 1e-12 < rel_err(res.P_new, ref["P_new"])      # it IS the reduced-precision path
    eng32.load(prob)                                     # resident path: bitwise reproducible
    eng32.run(); r1 = eng32.result()
    eng32.run(); r2 = eng32.result()
    assert np.array_equal(r1.dx, r2.dx) and np.array_equal(r1.P_new, r2.P_new)


def test_config5_full_size_f32():
    """BASELINE.json configs[4] as written: N = 50, 20000 features, track 15, fp32 storage + f32 MFMA P-update."""
    from msckf_amd.api import UpdateEngine
    from test_gpu_parity import _big_case
    prob, ref = _big_case(50, 20000, 15)
    with UpdateEngine(max_clones=50, max_features=20000, max_track=15, dtype="f32") as e:
        res = e.update_problem(prob)
        assert res.status == 0
        assert np.array_equal(res.accepted, ref["accepted"])
        assert rel_err(res.dx, ref["dx"]) < TOL_DX and rel_err(res.P_new, ref["P_new"]) < TOL_P


def test_bad_dtype_is_refused():
    from msckf_amd.api import UpdateEngine
    with pytest.raises(ValueError):
        UpdateEngine(dtype="bf16")


def test_ragged_long_tracks_keep_the_tolerance(eng32):
    """A ragged batch with split long tracks -- tens of dense remainder row blocks through K6-K7 -- on a 48-clone
    window: the batch `tools/soak_holes.py 150 8 f32` found 2.5e-4 off on dx while the P-update's rank-16 products of those blocks
    ran on the f32 matrix cores; with split long tracks in the batch they stay fp64 (6e-6).  reference MSCKF.py:604-614."""
    import importlib.util, os
    from oracle import msckf_oracle as oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak_holes", os.path.join(root, "tools", "soak_holes.py"))
    sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
    rng = np.random.default_rng(8)
    for _ in range(124):                                   # (the soak's own sequence: case 123)
        N = int(rng.integers(2, 54)); F = int(rng.integers(1, 400))
        hi = int(rng.integers(2, min(N, 31) + 1))
        prob = sh.ragged(rng, N, F, 2, hi, float(rng.choice([0.0, 0.1, 0.4])))
    assert (prob.N, prob.F) == (48, 370)
    ref = oracle.update(prob, dense_noise=False)
    res = eng32.update_problem(prob)
    assert res.status == ref["status"] == 0 and np.array_equal(res.accepted, ref["accepted"])
    assert eng32.debug_split()["long_tracks"] > 0
    assert rel_err(res.dx, ref["dx"]) < TOL_DX and rel_err(res.P_new, ref["P_new"]) < TOL_P
