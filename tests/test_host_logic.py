"""CPU tests of the host-side mirror: packing of reference-shaped objects, state
injection (reference MSCKF.correct :616-661), chi-square table, sharding."""
import os
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, load_golden, load_sequence, rel_err, sequence_cases


def make_reference_shaped(prob, ref):
    """Mock objects with the attributes MSCKF.update / correct touch."""
    keys = [7 * (i + 3) for i in range(prob.N)]
    cams = OrderedDict()
    for i, k in enumerate(keys):
        pose = SimpleNamespace(R=prob.cam_R[i].copy(), t=prob.cam_t[i].copy())
        same = np.array_equal(prob.cam_R0[i], prob.cam_R[i]) and np.array_equal(prob.cam_t0[i], prob.cam_t[i])
        null = pose if same else SimpleNamespace(R=prob.cam_R0[i].copy(), t=prob.cam_t0[i].copy())
        cams[k] = SimpleNamespace(T_W_Ci=pose, T_W_Ci_null=null)
    imu = SimpleNamespace(W_gravity=prob.gravity.copy(),
                          T_W_Ii=SimpleNamespace(R=ref["imu_R"].copy(), t=ref["imu_t"].copy()),
                          v_W_Ii=ref["imu_v"].copy(), gyroscope_bias=ref["imu_bg"].copy(),
                          accelerometer_bias=ref["imu_ba"].copy())
    state = SimpleNamespace(cameras=cams, covariance=prob.P.copy(), imu=imu)
    filt = SimpleNamespace(state=state, K=prob.K, sigma_image=prob.sigma,
                           number_of_residuals_discarded_for_gasting_test=0)
    feats = OrderedDict()
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        idp = SimpleNamespace(base=prob.idp_base[j].copy(), m=prob.idp_m[j].copy(), rho=float(prob.idp_rho[j]))
        feats[1000 + j] = SimpleNamespace(keypoints=[prob.obs_uv[i].copy() for i in range(a, b)],
                                          camera_indices=[keys[int(prob.obs_slot[i])] for i in range(a, b)],
                                          inverse_depth_point=idp)
    return filt, feats


@pytest.mark.parametrize("case", ["cfg1_A", "edge_null_pose", "edge_variable_tracks"])
def test_pack_roundtrip(case):
    from msckf_amd.pack import problem_from_reference
    prob, ref = load_golden(case)
    filt, feats = make_reference_shaped(prob, ref)
    p2 = problem_from_reference(filt, feats)
    for name in ["P", "cam_R", "cam_t", "cam_R0", "cam_t0", "gravity", "view_ptr", "obs_uv", "obs_slot",
                 "idp_base", "idp_m", "idp_rho"]:
        assert np.array_equal(getattr(p2, name), getattr(prob, name)), name
    assert p2.sigma == prob.sigma


@pytest.mark.parametrize("case", ["cfg1_A", "cfg1_B", "edge_null_pose", "edge_some_rejected"])
def test_inject_state_matches_reference(case):
    from msckf_amd.inject import inject_state
    prob, ref = load_golden(case)
    filt, _ = make_reference_shaped(prob, ref)
    inject_state(filt.state, ref["dx"])
    np.testing.assert_allclose(filt.state.imu.T_W_Ii.R, ref["post_imu_R"], atol=1e-13)
    np.testing.assert_allclose(filt.state.imu.T_W_Ii.t, ref["post_imu_t"], atol=1e-13)
    np.testing.assert_allclose(filt.state.imu.v_W_Ii, ref["post_imu_v"], atol=1e-13)
    np.testing.assert_allclose(filt.state.imu.gyroscope_bias, ref["post_imu_bg"], atol=1e-13)
    np.testing.assert_allclose(filt.state.imu.accelerometer_bias, ref["post_imu_ba"], atol=1e-13)
    for i, cam in enumerate(filt.state.cameras.values()):
        np.testing.assert_allclose(cam.T_W_Ci.R, ref["post_cam_R"][i], atol=1e-13)
        np.testing.assert_allclose(cam.T_W_Ci.t, ref["post_cam_t"][i], atol=1e-13)


def test_packaged_chi2_table_equals_golden():
    from msckf_amd.api import chi2_table
    t = np.load(os.path.join(GOLDEN_DIR, "chi2_ppf_095.npy"))
    assert np.array_equal(chi2_table(), t)


def test_synth_is_seeded():
    from msckf_amd import synth
    a = synth.make_problem(10, 30, 5, seed=3)
    b = synth.make_problem(10, 30, 5, seed=3)
    c = synth.make_problem(10, 30, 5, seed=4)
    assert np.array_equal(a.obs_uv, b.obs_uv) and np.array_equal(a.P, b.P)
    assert not np.array_equal(a.obs_uv, c.obs_uv)
    assert a.view_ptr[-1] == a.obs_slot.size == 150


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_covers_and_balances(world):
    from msckf_amd import synth
    from msckf_amd.shard import partition_features
    p = synth.make_problem(12, 200, 10, seed=1, variable_tracks=True)
    parts = partition_features(p.view_ptr, world)
    assert parts[0][0] == 0 and parts[-1][1] == p.F
    for (a, b), (c, d) in zip(parts[:-1], parts[1:]):
        assert b == c and a <= b
    rows = [2 * int(p.view_ptr[hi] - p.view_ptr[lo]) for lo, hi in parts]
    assert max(rows) - min(rows) <= 2 * 2 * 10 + 2


def test_subset_shares_state():
    from msckf_amd import synth
    p = synth.make_problem(8, 40, 5, seed=2)
    s = p.subset(10, 25)
    assert s.F == 15 and s.view_ptr[0] == 0 and s.P is p.P
    assert np.array_equal(s.obs_uv, p.obs_uv[p.view_ptr[10]:p.view_ptr[25]])


def test_take_matches_subset_and_reorders():
    from msckf_amd import synth
    prob = synth.make_problem(8, 20, 5, seed=2, variable_tracks=True)
    a, b = prob.take(np.arange(3, 9)), prob.subset(3, 9)
    for name in ("view_ptr", "obs_uv", "obs_slot", "idp_base", "idp_m", "idp_rho"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    r = prob.take([7, 2])
    assert np.array_equal(r.idp_rho, prob.idp_rho[[7, 2]])
    assert np.array_equal(r.obs_uv[: r.view_ptr[1]], prob.obs_uv[prob.view_ptr[7]:prob.view_ptr[8]])
    assert prob.take([]).F == 0


def test_tracks_pack_roundtrip():
    """Reference-shaped Feature.lines / counters -> TrackTable (pack.tracks_from_reference)."""
    from collections import OrderedDict as OD
    from conftest import load_golden_select
    from msckf_amd.pack import select_params_from_reference, tracks_from_reference
    prob, tracks, params, _ = load_golden_select("sel_variable_tracks")
    feats = OD()
    for j in range(prob.F):
        a, b = int(prob.view_ptr[j]), int(prob.view_ptr[j + 1])
        feats[j] = SimpleNamespace(
            keypoints=[None] * (b - a),
            lines=[SimpleNamespace(base=tracks.line_base[i], direction=tracks.line_dir[i].reshape(3, 1),
                                   confidence=tracks.line_conf[i]) for i in range(a, b)],
            lost_for_n_frames=int(tracks.lost_for[j]), tracked_for_n_frames=int(tracks.tracked_for[j]))
    got = tracks_from_reference(feats)
    for name in ("line_base", "line_dir", "line_conf", "lost_for", "tracked_for"):
        assert np.array_equal(getattr(got, name), getattr(tracks, name)), name
    feats[0].lines.pop()
    with pytest.raises(ValueError):
        tracks_from_reference(feats)
    cam = SimpleNamespace(width=320, height=200)
    filt = SimpleNamespace(state=SimpleNamespace(cameras=OD([(1, cam)])), min_number_of_frames_to_be_lost=1,
                           min_number_of_frames_to_be_tracked=5, use_parallax=True, min_parallax=20)
    sp = select_params_from_reference(filt)
    assert (sp.width, sp.height, sp.min_frames_tracked, sp.min_parallax_deg) == (320, 200, 5, 20.0)


def test_make_tracks_is_seeded():
    from msckf_amd import synth
    prob = synth.make_problem(8, 20, 5, seed=2)
    a, b = synth.make_tracks(prob, 3, flip_fraction=0.2), synth.make_tracks(prob, 3, flip_fraction=0.2)
    assert np.array_equal(a.line_dir, b.line_dir) and np.array_equal(a.lost_for, b.lost_for)
    assert not np.array_equal(a.line_conf, synth.make_tracks(prob, 4).line_conf)


@pytest.mark.parametrize("case", sequence_cases())
def test_host_transition_and_augmentation_match_reference(case):
    """propagation.py (the host set-up of msckf_propagate / msckf_augment) against every
    process_imu / state_augmentation step of the reference run: the 15x15 Phi, Q and the 6x15 J
    are pinned through the covariance the reference left after the step."""
    from msckf_amd import propagation
    head, ops = load_sequence(case)
    P = head["P0"]
    for op in ops:
        if op["kind"] == 0:
            Phi, Q = propagation.imu_transition(op["R"], op["t"], op["v"], op["R0"], op["t0"], op["v0"], op["gyro"],
                                                op["acc"], float(op["dt"]), head["gravity"], head["Qc"], op["w_planet"])
            Pn = P.copy()
            Pn[:15, :15] = Phi @ P[:15, :15] @ Phi.T + Q
            Pn[:15, 15:] = Phi @ P[:15, 15:]
            Pn[15:, :15] = Pn[:15, 15:].T
            assert rel_err((Pn + Pn.T) / 2, op["P_after"]) < 1e-13
        elif op["kind"] == 1:
            J, cR, ct = propagation.augmentation(op["imu_R"], op["imu_t"], (head["T_W_I_R"], head["T_W_I_t"]),
                                                 (head["T_W_C_R"], head["T_W_C_t"]))
            np.testing.assert_allclose(cR, op["cam_R"], atol=1e-14)
            np.testing.assert_allclose(ct, op["cam_t"], atol=1e-14)
            d = P.shape[0]
            M = np.vstack([np.eye(d), np.hstack([J, np.zeros((6, d - 15))])])
            S = M @ P @ M.T
            assert rel_err((S + S.T) / 2, op["P_after"]) < 1e-14
        P = op["P_after"]


def test_max_span_of_a_batch():
    """`UpdateEngine.max_span` (host side, no GPU): the longest track of the whole batch in clone slots is what
    every rank passes to the library's `msckf_band_rule` before sharding (the rule itself lives in the library
    and is tested on the GPU, tests/test_gpu_parity.py::test_band_rule_comes_from_the_library)."""
    from msckf_amd import synth
    from msckf_amd.api import UpdateEngine
    assert UpdateEngine.max_span(synth.make_problem(30, 200, 10, seed=1)) == 10
    assert UpdateEngine.max_span(synth.make_problem(12, 40, 10, seed=2, variable_tracks=True)) <= 10
    assert UpdateEngine.max_span(synth.make_problem(16, 40, 14, seed=3)) == 14
    wide = synth.make_problem(30, 50, 10, seed=5)
    slots = wide.obs_slot.copy()
    a, b = int(wide.view_ptr[3]), int(wide.view_ptr[4])
    slots[b - 1] = min(int(slots[a]) + 12, 29)                                       # one track skips ahead: span 13
    wide.obs_slot = slots
    assert UpdateEngine.max_span(wide) == 13
    assert UpdateEngine.max_span(synth.make_problem(5, 0, 3, seed=0)) == 0


# ---- the RCCL driver's host logic on the CPU (the engine and the fabric are stand-ins, the compute is the oracle) ----

class _Fabric:
    def __init__(self):
        self.records, self.result, self.log = {}, None, []


class _FakeEngine:
    """The calls `shard.RcclShardedUpdate` makes on an engine, recorded and answered from the oracle."""

    def __init__(self, rank, world, fabric):
        self.rank, self.world, self.fab = rank, world, fabric
        self.shard = self.bounds = self.span = None
        self.xchg = False

    @staticmethod
    def max_span(prob):
        from msckf_amd.api import UpdateEngine
        return UpdateEngine.max_span(prob)

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, rank, world, uid):
        assert (rank, world) == (self.rank, self.world) and len(uid) == 128

    def band_ok(self, prob):
        return self.max_span(prob) <= 15

    def set_group_exchange(self, on):
        self.xchg = bool(on)

    def set_exchange_span(self, span):
        self.span = int(span)

    def set_exchange_mask(self, bounds):
        self.bounds = np.asarray(bounds).copy()

    def load(self, prob):
        assert self.bounds is not None and self.span is not None      # layout first, then the shard
        self.shard = prob
        self.n_clones = prob.N

    def group_record_doubles(self):
        return 1000 + self.shard.N

    def comm_buffer(self, n):
        assert n == self.group_record_doubles() * (self.world + 1) + 8
        return 1 << 20

    def run_compress(self):
        self.fab.log.append(("compress", self.rank))

    def device_pointer(self, which):
        return 100 + which

    def comm_gather(self, send, recv, count, root):
        assert send == 103 and count == self.group_record_doubles() and root == 0
        self.fab.records[self.rank] = self.shard

    def merge_groups_flags(self, recv, n_rec, flags):
        assert self.rank == 0 and n_rec == self.world and flags.shape == (self.world, self.shard.N)
        from oracle import msckf_oracle as oracle
        parts = [self.fab.records[r] for r in range(self.world)]          # every rank has deposited its record
        vp = np.concatenate([[0]] + [p.view_ptr[1:] + sum(int(q.view_ptr[-1]) for q in parts[:i]) for i, p in enumerate(parts)])
        for r, p in enumerate(parts):                                        # the flags say which groups a record carries
            first = np.unique(np.minimum.reduceat(p.obs_slot, p.view_ptr[:-1])) if p.F else np.zeros(0, dtype=int)
            assert np.array_equal(np.nonzero(flags[r])[0], first)
        full = parts[0].__class__(
            P=parts[0].P, cam_R=parts[0].cam_R, cam_t=parts[0].cam_t, cam_R0=parts[0].cam_R0, cam_t0=parts[0].cam_t0,
            gravity=parts[0].gravity, K=parts[0].K, sigma=parts[0].sigma, view_ptr=vp.astype(np.int32),
            obs_uv=np.concatenate([p.obs_uv for p in parts]), obs_slot=np.concatenate([p.obs_slot for p in parts]),
            idp_base=np.concatenate([p.idp_base for p in parts]), idp_m=np.concatenate([p.idp_m for p in parts]),
            idp_rho=np.concatenate([p.idp_rho for p in parts]))
        self.fab.result = oracle.update(full, dense_noise=False)

    def result_range_doubles(self):
        d = 15 + 6 * self.shard.N
        return 8 + d + d * d + (int(self.bounds[-1]) + 7) // 8

    def comm_broadcast(self, ptr, count, root):
        assert ptr == 105 and count == self.result_range_doubles() and root == 0
        self.fab.log.append(("bcast", self.rank))

    def shared_result(self):
        from msckf_amd.api import UpdateResult
        r = self.fab.result
        return UpdateResult(int(r["status"]), r["dx"], r["P_new"], r["accepted"].astype(np.uint8),
                            {"n_rejected": int(len(r["accepted"]) - r["accepted"].sum())})


@pytest.mark.parametrize("world,kw", [(3, {"outlier_fraction": 0.1, "outlier_px": 400.0}), (5, {"variable_tracks": True})])
def test_rccl_driver_host_logic(world, kw):
    """`RcclShardedUpdate` at world > 1 on the CPU: every rank configures the same record layout before it loads its
    shard, the shards tile the batch, the flags match the shards' groups, only rank 0 merges, every rank broadcasts the
    same range and reads the same (status, dx, P+, accepted, rejected) -- the oracle's result on the whole batch."""
    from msckf_amd import synth
    from msckf_amd.shard import RcclShardedUpdate
    from oracle import msckf_oracle as oracle
    prob = synth.make_problem(12, 90, 8, seed=5, **kw)
    ref = oracle.update(prob, dense_noise=False)
    fab = _Fabric()
    drv = [RcclShardedUpdate(_FakeEngine(r, world, fab), r, world, bytes(128)) for r in range(world)]
    for dv in drv:
        dv.load(prob)
    assert all(np.array_equal(dv.bounds, drv[0].bounds) for dv in drv) and drv[0].bounds[-1] == prob.F
    assert sum(dv.shard[1] - dv.shard[0] for dv in drv) == prob.F and all(dv.groups for dv in drv)
    for r in list(range(1, world)) + [0]:            # the root's merge needs every record: run it last in this serial stand-in
        drv[r].step()
    outs = [dv.result() for dv in drv]
    for st, dx, P, acc, nrej in outs:
        assert st == ref["status"] and np.array_equal(acc, ref["accepted"]) and nrej == prob.F - int(ref["accepted"].sum())
        assert rel_err(dx, ref["dx"]) < 1e-12 and rel_err(P, ref["P_new"]) < 1e-12
    assert [x for x in fab.log if x[0] == "bcast"] == [("bcast", r) for r in list(range(1, world)) + [0]]


def test_unique_id_file_carries_the_launch_tag(tmp_path):
    """A file left by another launch (other tag) is ignored; this launch's file is read back; rank 0 removes it."""
    from msckf_amd.shard import RcclShardedUpdate, exchange_unique_id
    fab = _Fabric()
    e0, e1 = _FakeEngine(0, 2, fab), _FakeEngine(1, 2, fab)
    path = str(tmp_path / "id")
    with open(path, "wb") as fh:
        fh.write(b"\x07" * 16 + b"\x09" * 128)
    with pytest.raises(TimeoutError):
        exchange_unique_id(e1, 1, 2, path, timeout_s=0.1)
    uid = exchange_unique_id(e0, 0, 2, path)
    assert exchange_unique_id(e1, 1, 2, path) == uid == bytes(range(128))
    RcclShardedUpdate(e0, 0, 2, uid, id_path=path)
    assert not os.path.exists(path)


def test_generated_sweep_groups_header_is_current():
    """csrc/sweep_dpp_groups.h is generated (tools/gen_sweep_dpp_groups.py): the committed file is what the generator
    writes, every statement stays within the 30 operands inline assembly allows, opens with the one s_nop 1 that covers
    the VALU-write -> DPP-read hazard, and rewrites the broadcast register last in its row."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_sweep_dpp_groups", os.path.join(ROOT, "tools", "gen_sweep_dpp_groups.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = gen.render()
    with open(gen.PATH) as f:
        assert f.read() == text
    stmts = re.findall(r'asm volatile\("(.*?)"\s*\n\s*: (.*?)\n\s*: (.*?)\);', text, flags=re.S)
    assert len(stmts) == sum(gen.gmax_dots(ns) + gen.gmax_update(ns) for ns in range(1, 7))
    for body, outs, ins in stmts:
        n_ops = 2 * outs.count('"+v"') + ins.count('"v"') + ins.count('"i"')
        assert n_ops <= 30
        lines = body.split("\\n\\t")
        assert lines[0] == "s_nop 1" and all(l.startswith("v_fmac_f64_dpp ") for l in lines[1:])
    for ns in range(1, 7):
        for g in range(1, gen.gmax_update(ns) + 1):
            body = re.search(r'struct DppUpdate<%d, %d> \{.*?asm volatile\("(.*?)"' % (ns, g), text, flags=re.S).group(1)
            lines = body.split("\\n\\t")[1:]
            for r in range(g):
                row = lines[r * ns:(r + 1) * ns]
                bc = "%%%d" % (r * ns)                                   # the row's broadcast register (slot K0)
                assert all(l.split()[2].rstrip(",") == bc for l in row)  # every FMA of the row reads it through DPP
                assert [l.split()[1].rstrip(",") == bc for l in row] == [False] * (ns - 1) + [True]   # and only the last writes it


def test_generated_feature_groups_header_is_current():
    """csrc/feature_dpp_groups.h (tools/gen_feature_dpp_groups.py): committed file == generator output, <= 30 operands per
    statement, one s_nop 1 in front, and the broadcast (register, lane) of every FMA addresses the table entry it stands for."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_feature_dpp_groups", os.path.join(ROOT, "tools", "gen_feature_dpp_groups.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = gen.render()
    with open(gen.PATH) as f:
        assert f.read() == text
    blocks = re.findall(r'struct (FeatGateView|FeatCorrRows)<(\d+)> \{.*?asm volatile\("(.*?)"\s*\n\s*: (.*?)\n\s*: (.*?)\);', text, flags=re.S)
    assert len(blocks) == gen.MAXV + 8
    for kind, num, body, outs, ins in blocks:
        num = int(num)
        in_list = [x.strip() for x in ins.split('", ')]
        in_list = re.findall(r'"v"\((.*?)\)(?:,|$)', ins)
        n_out = outs.count('"+v"')
        assert 2 * n_out + len(in_list) <= 30
        lines = body.split("\\n\\t")
        assert lines[0] == "s_nop 1"
        k = 0
        for l in lines[1:]:
            m = re.match(r"v_fmac_f64_dpp %(\d+), %(\d+), (-?)%(\d+) row_newbcast:(\d+) ", l)
            acc, src, neg, oth, lane = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4)), int(m.group(5))
            name = in_list[src - n_out]
            reg = int(re.findall(r"\[(\d+)\]", name)[-1])
            entry = 16 * reg + lane
            if kind == "FeatGateView":
                a, which = divmod(k, 5)
                if which < 3:
                    assert name.startswith("zq[%d]" % which) and entry == 6 * num + a and acc == which
                else:
                    assert name.startswith("aq[") and entry == (2 * num + which - 3) * 6 + a and acc == which
                assert in_list[oth - n_out] == "pv[%d]" % a and neg == ""
                k += 1
            else:
                i, t = acc, oth - n_out
                assert name.startswith("vq[") and entry == (num + i) * 3 + t and neg == "-" and in_list[oth - n_out] == "w%d" % t


def test_concat_problems_and_few_long_tracks():
    """synth.concat_problems / few_long_tracks_problem (the mixed-span batches of bench.py and tests): a valid CSR batch on one
    state, and the oracle's update of it is the update of its parts' stacked rows (order independence, MSCKF.py:573)."""
    import msckf_amd  # noqa: F401
    from msckf_amd import synth
    from oracle import msckf_oracle as oracle
    a = synth.make_problem(12, 40, 5, seed=3)
    b = synth.make_problem(12, 4, 12, seed=4, P=a.P, poses=(a.cam_R, a.cam_t))
    c = synth.concat_problems(a, b)
    assert c.F == 44 and c.view_ptr[0] == 0 and np.all(np.diff(c.view_ptr) >= 1)
    assert c.view_ptr[-1] == len(c.obs_slot) == len(c.obs_uv)
    assert np.array_equal(c.P, a.P) and np.array_equal(c.cam_t, a.cam_t)
    out_c = oracle.update(c, dense_noise=False)
    swapped = synth.concat_problems(b, synth.UpdateProblem(**{**a.__dict__}))
    out_s = oracle.update(swapped, dense_noise=False)
    assert out_c["status"] == out_s["status"] == 0
    assert np.linalg.norm(out_c["dx"] - out_s["dx"]) < 1e-10 * np.linalg.norm(out_c["dx"])
    assert np.linalg.norm(out_c["P_new"] - out_s["P_new"]) < 1e-11 * np.linalg.norm(out_c["P_new"])
    p = synth.few_long_tracks_problem(16, 60, 3, 6, seed=1)
    lens = np.diff(p.view_ptr)
    assert p.F == 60 and (lens == 16).sum() == 3 and (lens == 6).sum() == 57


def test_plan_flags_of_the_header_and_the_binding_agree():
    """MSCKF_FLAG_* in include/msckf_mi355x.h == the values monocular-visual-inertial-msckf_amd/_ffi.py passes (plan="band")."""
    import re
    import msckf_amd  # noqa: F401
    from msckf_amd import _ffi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "msckf_mi355x.h")).read()
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (MSCKF_FLAG_\w+) (\d+)", hdr)}
    assert vals["MSCKF_FLAG_TREE_PLAN"] == _ffi.FLAG_TREE_PLAN == 1
    assert vals["MSCKF_FLAG_BAND_ONLY"] == _ffi.FLAG_BAND_ONLY == 2
