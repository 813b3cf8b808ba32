"""The host side of the drop-in call (round 4): tracks uploaded in the caller's order and permuted on the device (k_gather),
state and K5 plan on a side stream, results mirrored into pinned host memory by the kernels.  None of it may change a bit of
the result: the one-shot call is compared with the resident sequence (which copies results the old way) and with itself
under every switch, and the error paths are exercised while uploads are in flight."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-8


def _engine(monkeypatch, **env):
    from msckf_amd.api import UpdateEngine
    for k in ("MSCKF_DIRECT_RESULT", "MSCKF_ZEROCOPY_MAX", "MSCKF_HOST_THREADS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return UpdateEngine(max_clones=30, max_features=12000, max_track=31)


def _shuffled(prob, seed):
    """The same batch with its tracks in a random order (the caller's order is arbitrary: a dict's insertion order)."""
    from msckf_amd import synth
    rng = np.random.default_rng(seed)
    order = rng.permutation(prob.F)
    M = np.diff(prob.view_ptr)
    vp = np.zeros(prob.F + 1, dtype=np.int32)
    vp[1:] = np.cumsum(M[order])
    idx = np.concatenate([np.arange(prob.view_ptr[f], prob.view_ptr[f + 1]) for f in order])
    q = synth.UpdateProblem(**{**prob.__dict__})
    q.view_ptr = vp
    q.obs_uv = prob.obs_uv[idx].copy(); q.obs_slot = prob.obs_slot[idx].copy()
    q.idp_base = prob.idp_base[order].copy(); q.idp_m = prob.idp_m[order].copy(); q.idp_rho = prob.idp_rho[order].copy()
    return q, order


@pytest.mark.parametrize("N,F,M,kw", [
    (30, 2000, 10, {}),                                                    # zero-copy tables, fused K6-K7, mirrored results
    (30, 6000, 10, dict(outlier_fraction=0.2, outlier_px=300.0)),           # above the zero-copy size: DMA
    (30, 1500, 10, dict(variable_tracks=True, outlier_fraction=0.1, outlier_px=300.0)),
    (12, 300, 5, {}),                                                      # below the host pool's size
])
def test_one_shot_equals_resident_under_every_switch(monkeypatch, N, F, M, kw):
    from msckf_amd import synth
    prob = synth.make_problem(N, F, M, seed=7, **kw)
    prob, _ = _shuffled(prob, 3)
    with _engine(monkeypatch) as eng:
        eng.load(prob)
        eng.run()
        ref = eng.result()                                                 # results copied back by hipMemcpyAsync
        assert ref.status == 0 and 0 < ref.accepted.sum() <= F
        one = eng.update_problem(prob)
        again = eng.update_problem(prob)
    for r in (one, again):
        assert r.status == 0
        assert np.array_equal(r.dx, ref.dx) and np.array_equal(r.P_new, ref.P_new) and np.array_equal(r.accepted, ref.accepted)
        assert r.stats["n_accepted"] == ref.stats["n_accepted"] and r.stats["stacked_rows"] == ref.stats["stacked_rows"]
        assert r.n_rejected == ref.n_rejected
    for env in (dict(MSCKF_DIRECT_RESULT=0), dict(MSCKF_ZEROCOPY_MAX=0), dict(MSCKF_ZEROCOPY_MAX=100000), dict(MSCKF_HOST_THREADS=0)):
        with _engine(monkeypatch, **env) as eng:
            r = eng.update_problem(prob)
        assert np.array_equal(r.dx, ref.dx) and np.array_equal(r.P_new, ref.P_new) and np.array_equal(r.accepted, ref.accepted), env


def test_track_order_does_not_matter(monkeypatch):
    """k_gather's permutation against the host's: the sorted image is the same whatever the caller's order, so dx / P+ are
    bitwise those of the unshuffled batch and the mask comes back in the caller's order."""
    from msckf_amd import synth
    prob = synth.make_problem(30, 3000, 10, seed=11, variable_tracks=True, outlier_fraction=0.15, outlier_px=300.0)
    with _engine(monkeypatch) as eng:
        base = eng.update_problem(prob)
        for seed in (1, 2):
            q, order = _shuffled(prob, seed)
            r = eng.update_problem(q)
            assert np.array_equal(r.accepted, base.accepted[order])
            # (tracks with equal first / last slot keep the caller's relative order: the sums differ in the last bits)
            assert rel_err(r.dx, base.dx) < 1e-10 and rel_err(r.P_new, base.P_new) < 1e-10


def test_golden_through_the_mirrored_results(monkeypatch):
    with _engine(monkeypatch) as eng:
        for case in ("cfg1_A", "cfg2_A", "cfg2_B", "edge_variable_tracks", "edge_all_rejected"):
            try:
                prob, ref = load_golden(case)
            except FileNotFoundError:
                continue
            res = eng.update_problem(prob)
            assert res.status == int(ref["status"]), case
            assert np.array_equal(res.accepted, ref["accepted"]), case
            assert rel_err(res.dx, ref["dx"]) < TOL and rel_err(res.P_new, ref["P_new"]) < TOL, case


def test_commit_and_resident_run_behind_a_one_shot_call(monkeypatch):
    """P+ of the one-shot call is in HBM as well as in the host mirror: commit it and run the next batch on it."""
    from msckf_amd import synth
    p1 = synth.make_problem(30, 2000, 10, seed=21)
    p2 = synth.make_problem(30, 2000, 10, seed=22)
    with _engine(monkeypatch) as eng:
        r1 = eng.update_problem(p1)
        assert eng.commit_covariance() == 0
        assert np.array_equal(eng.covariance(), r1.P_new)
        eng.set_poses(p2.cam_R, p2.cam_t, p2.cam_R0, p2.cam_t0)
        eng.set_features(p2)
        eng.run()
        res = eng.result()                                   # the copy path again, behind a mirrored run
        q = synth.UpdateProblem(**{**p2.__dict__})
        q.P = r1.P_new
        q.gravity = p1.gravity; q.K = p1.K; q.sigma = p1.sigma
        chk = eng.update_problem(q)
    assert res.status == 0
    assert np.array_equal(res.dx, chk.dx) and np.array_equal(res.P_new, chk.P_new)


def test_errors_while_uploads_are_in_flight(monkeypatch):
    """A slot out of range / a duplicate slot is found AFTER the observations have started up by DMA (and the state on the side
    stream): the call returns the code, the engine drains its streams and the next call is clean."""
    from msckf_amd import synth
    from msckf_amd._ffi import EngineError, ERR_ARG, ERR_DUP_SLOT
    good = synth.make_problem(30, 5000, 10, seed=31)
    with _engine(monkeypatch) as eng:
        ref = eng.update_problem(good)
        for where, val, code in ((40000, 30, ERR_ARG), (49999, -1, ERR_ARG), (123, None, ERR_DUP_SLOT)):
            bad = synth.UpdateProblem(**{**good.__dict__})
            sl = good.obs_slot.copy()
            sl[where] = sl[where + 1] if val is None else val
            bad.obs_slot = sl
            with pytest.raises(EngineError) as ei:
                eng.update_problem(bad)
            assert ei.value.code == code, (where, val)
            r = eng.update_problem(good)
            assert np.array_equal(r.dx, ref.dx) and np.array_equal(r.P_new, ref.P_new)


def test_no_accepted_track_and_empty_batch(monkeypatch):
    """The reference's early returns (MSCKF.py:584-585, 591-592) through the mirrored results: status 1, dx = 0, P untouched."""
    from msckf_amd import synth
    prob, ref = load_golden("edge_all_rejected")
    with _engine(monkeypatch) as eng:
        r = eng.update_problem(prob)
        assert r.status == 1 and r.accepted.sum() == 0
        assert np.array_equal(r.P_new, prob.P) and not r.dx.any()
        assert r.n_rejected == int(ref["n_rejected"])
        good = synth.make_problem(20, 1500, 8, seed=6)
        assert eng.update_problem(good).status == 0
        empty = synth.UpdateProblem(**{**good.__dict__})
        empty.view_ptr = np.zeros(1, dtype=np.int32)
        empty.obs_uv = np.zeros((0, 2)); empty.obs_slot = np.zeros(0, dtype=np.int32)
        empty.idp_base = np.zeros((0, 3)); empty.idp_m = np.zeros((0, 3)); empty.idp_rho = np.zeros(0)
        r = eng.update_problem(empty)
        assert r.status == 1 and np.array_equal(r.P_new, good.P) and not r.dx.any()
        assert eng.update_problem(good).status == 0


def test_resident_calls_without_waits_keep_their_order(monkeypatch):
    """msckf_set_features / msckf_set_poses / msckf_commit_covariance / msckf_run leave their work in the stream; a call that
    rewrites a pinned staging buffer or uses the side stream waits first.  Back-to-back sequences nobody waited for must give
    the results of the same calls made one by one with a wait after each."""
    from msckf_amd import synth
    pa = synth.make_problem(30, 2000, 10, seed=41)
    pb = synth.make_problem(30, 3000, 8, seed=42, variable_tracks=True)
    pc = synth.make_problem(30, 1800, 10, seed=43)
    with _engine(monkeypatch) as eng:
        ref_a = eng.update_problem(pa)
        qb = synth.UpdateProblem(**{**pb.__dict__}); qb.P = ref_a.P_new
        qb.gravity = pa.gravity; qb.K = pa.K; qb.sigma = pa.sigma
        ref_b = eng.update_problem(qb)                        # pb's tracks and poses on pa's P+
        qc = synth.UpdateProblem(**{**pc.__dict__}); qc.P = ref_b.P_new
        qc.gravity = pa.gravity; qc.K = pa.K; qc.sigma = pa.sigma
        ref_c = eng.update_problem(qc)

        eng.set_state(pa)
        eng.set_features(pc)                                  # overwritten before anyone waited for it
        eng.set_features(pa)
        eng.run()
        eng.set_features(pa)                                  # a run is pending: the plan must not overtake it on the side stream
        eng.run()
        ra = eng.result()
        assert np.array_equal(ra.dx, ref_a.dx) and np.array_equal(ra.P_new, ref_a.P_new)
        assert eng.commit_covariance() == 0
        eng.set_poses(pc.cam_R, pc.cam_t, pc.cam_R0, pc.cam_t0)
        eng.set_poses(pb.cam_R, pb.cam_t, pb.cam_R0, pb.cam_t0)          # the pose staging buffer twice in a row
        eng.set_features(pb)
        eng.run()
        rb = eng.result()
        assert np.array_equal(rb.dx, ref_b.dx) and np.array_equal(rb.P_new, ref_b.P_new)
        assert np.array_equal(rb.accepted, ref_b.accepted)
        assert eng.commit_covariance() == 0                   # left in the stream ...
        rc = eng.update_problem(qc)                           # ... and the one-shot call's side stream waits for it
        assert np.array_equal(rc.dx, ref_c.dx) and np.array_equal(rc.P_new, ref_c.P_new)
        assert np.array_equal(eng.covariance(), ref_b.P_new)  # (the one-shot call does not commit)


@pytest.mark.parametrize("N,F,M,kw", [
    (30, 2000, 10, {}),                                   # one merge level, eight slots: k_root_gain with merge workgroups
    (30, 10000, 10, {}),                                  # 9 - 12 triangles per group: k_root_gain_m (eleven slots)
    (30, 3000, 10, dict(variable_tracks=True, outlier_fraction=0.1, outlier_px=300.0)),
    (12, 8000, 5, {}),                                    # large groups: two merge levels, only the last one is streamed
    (20, 500, 8, {}),
])
def test_streamed_merges_against_their_own_launch(monkeypatch, N, F, M, kw):
    """The last merge level inside the root's launch (rows taken as they are published, first triangle folded instead of
    adopted) against the same level in a launch of its own: the same R up to rounding (reference MSCKF.py:594-598 fixes
    T_H only up to the signs of its rows), so dx / P+ agree to 1e-10, and the streamed run is bitwise repeatable."""
    from msckf_amd import synth
    prob = synth.make_problem(N, F, M, seed=17, **kw)
    with _engine(monkeypatch, MSCKF_ROOT_STREAM=0) as eng:
        ref = eng.update_problem(prob)
    with _engine(monkeypatch) as eng:
        one = eng.update_problem(prob)
        assert one.status == ref.status == 0
        assert np.array_equal(one.accepted, ref.accepted)
        assert rel_err(one.dx, ref.dx) < 1e-10 and rel_err(one.P_new, ref.P_new) < 1e-10
        assert np.array_equal(one.P_new, one.P_new.T)
        eng.load(prob)
        for _ in range(12):                               # race screen: polling / publication order must not show in the result
            eng.run()
            r = eng.result()
            assert np.array_equal(r.dx, one.dx) and np.array_equal(r.P_new, one.P_new)


@pytest.mark.parametrize("N,F,M", [
    (5, 1, 3), (30, 2, 10), (30, 1023, 10), (30, 1024, 10),      # one track; either side of the host pool's size
    (30, 4096, 10), (30, 4097, 10),                               # either side of the copy-kernel / DMA switch
    (3, 40, 3), (30, 600, 30),                                    # a short window; tracks as long as the window allows
])
def test_upload_paths_at_their_boundaries(monkeypatch, N, F, M):
    """Sizes where the host path changes its route (pool on / off, k_stage / DMA, one wavefront of k_gather half empty):
    the one-shot call against the oracle (1e-8) and, bitwise, against the resident sequence."""
    from msckf_amd import synth
    from oracle import msckf_oracle as oracle
    prob = synth.make_problem(N, F, M, seed=100 + F, variable_tracks=M > 3, outlier_fraction=0.1 if F > 10 else 0.0, outlier_px=300.0)
    prob, _ = _shuffled(prob, F)
    ref = oracle.update(prob, dense_noise=False)
    with _engine(monkeypatch) as eng:
        one = eng.update_problem(prob)
        eng.load(prob)
        eng.run()
        res = eng.result()
    assert one.status == ref["status"] and np.array_equal(one.accepted, ref["accepted"])
    assert rel_err(one.dx, ref["dx"]) < TOL and rel_err(one.P_new, ref["P_new"]) < TOL
    assert np.array_equal(one.dx, res.dx) and np.array_equal(one.P_new, res.P_new) and np.array_equal(one.accepted, res.accepted)


def test_a_timed_out_fused_launch_is_retried_on_plain_launches():
    """The workgroups of k_root_gain wait for each other inside the launch; if one of them were kept off the device until the
    0.5 s bound (a partitioned or busy GPU) the update is run again on kernels that never wait inside a launch, and the context
    stays on those.  MSCKF_DEBUG_FAKE_TIMEOUT=1 makes the first update of a context read as timed out."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import msckf_amd
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from oracle import msckf_oracle as oracle
ok = True
with UpdateEngine(max_clones=30, max_features=600, max_track=10) as e:
    for seed in (1, 2, 3):
        p = synth.make_problem(30, 500, 10, seed=seed, outlier_fraction=0.05, outlier_px=300.0)
        ref = oracle.update(p, dense_noise=False)
        r = e.update_problem(p)
        e_dx = np.linalg.norm(r.dx - ref["dx"]) / np.linalg.norm(ref["dx"])
        e_P = np.linalg.norm(r.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"])
        ok = ok and r.status == 0 and np.array_equal(r.accepted, ref["accepted"]) and e_dx < 1e-8 and e_P < 1e-8
        print(seed, r.status, e_dx, e_P, r.stats.get("k5_launches"), flush=True)
print("RETRY_OK" if ok else "RETRY_FAIL")
""" % root
    env = dict(os.environ, MSCKF_DEBUG_FAKE_TIMEOUT="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "RETRY_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # the first call ran the fused launch (2 K5 launches), was retried, and every later call has the levels as launches of their own
    lines = [l.split() for l in out.stdout.splitlines() if l and l[0] in "123"]
    assert int(lines[0][4]) >= 3 and int(lines[1][4]) >= 3, lines
