#!/usr/bin/env python3
"""Rank-0 side of the sharded update on ONE GPU (no torch): time the merge of G shards' exports (already in HBM) +
K6-K7 for the two exchange formats (root blocks -> fold-tree merge, group records -> group folds + one root sweep),
and the local K1-K5 of a shard in both formats.  The group path is the shipped one (the calls of
RcclShardedUpdate.step on rank 0: merge_groups_flags on records lying in the exchange buffer).
usage: merge_bench.py [N F_total M]   default: configs[3] shapes (30, 2000 G, 10) and configs[4] (50, 20000, 15)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from msckf_amd.shard import partition_features, shard_group_flags


def run(N, F, M, G, dtype="f64"):
    prob = synth.make_problem(N, F, M, seed=0)
    shards = partition_features(prob.view_ptr, G)
    out = {}
    for mode in ("blocks", "groups"):
        with UpdateEngine(max_clones=N, max_features=F, max_track=M, dtype=dtype) as e:
            groups = mode == "groups"
            e.set_group_exchange(groups)
            e.set_exchange_span(e.max_span(prob) if groups else 0)
            count = None
            total = 0
            t_local = 0.0
            buf = 0
            for r, (lo, hi) in enumerate(shards):
                e.load(prob.subset(lo, hi))
                if count is None:
                    count = e.group_record_doubles() if groups else e.block_doubles()
                    buf = e.comm_buffer(count * G + 8)
                e.run_compress(); e.sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    e.run_compress()
                e.sync()
                t_local = max(t_local, (time.perf_counter() - t0) / 10 * 1e6)
                if groups:
                    e.export_groups(dst_ptr=buf + 8 * count * r, count=False)
                else:
                    _, n = e.export_block(dst_ptr=buf + 8 * count * r)
                    total += n
            e.set_state(prob)
            flags = shard_group_flags(prob, shards)
            merge = (lambda: e.merge_groups_flags(buf, G, flags)) if groups else (lambda: e.merge_gain(buf, total, n_blocks=G))
            for _ in range(3):
                merge(); e.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                merge(); e.sync()
            t_merge = (time.perf_counter() - t0) / 10 * 1e6
            res = e.result()
            out[mode] = (t_local, t_merge, res.dx, count * 8)
    d = np.linalg.norm(out["blocks"][2] - out["groups"][2]) / np.linalg.norm(out["blocks"][2])
    print(f"N={N} F={F} M={M} {dtype} G={G}: root blocks: local K1-K5 {out['blocks'][0]:.0f} us, merge + K6-K7 {out['blocks'][1]:.0f} us, "
          f"{out['blocks'][3] / 1e3:.0f} KB/rank | group records: local {out['groups'][0]:.0f} us, merge + K6-K7 {out['groups'][1]:.0f} us, "
          f"{out['groups'][3] / 1e3:.0f} KB/rank | dx diff {d:.1e}", flush=True)


if len(sys.argv) > 3:
    N, F, M = (int(x) for x in sys.argv[1:4])
    for G in (2, 4, 8):
        run(N, F, M, G)
else:
    for G in (2, 4, 8):
        run(30, 8000, 10, G)
    for G in (2, 8):
        run(50, 20000, 15, G)
    run(50, 20000, 15, 8, "f32")
