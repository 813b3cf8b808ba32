#!/usr/bin/env python3
"""Rank-0 side of the sharded update on ONE GPU: time the merge of G shards' exports (already in HBM) + K6-K7
for the two exchange formats (root blocks -> fold-tree merge, group triangles -> group folds + one root sweep),
and the local K1-K5 of a shard in both modes."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msckf_amd  # noqa: F401
from msckf_amd import synth
from msckf_amd.api import UpdateEngine
from msckf_amd.shard import partition_features
N, Fg, M = 30, 2000, 10
for G in (2, 4, 8):
    prob = synth.make_problem(N, Fg * G, M, seed=0)
    out = {}
    for mode in ("blocks", "groups"):
        with UpdateEngine(max_clones=N, max_features=Fg * G, max_track=M) as e:
            e.set_group_exchange(mode == "groups")
            payload, total = [], 0
            t_local = None
            for lo, hi in partition_features(prob.view_ptr, G):
                e.load(prob.subset(lo, hi))
                e.run_compress(); e.sync()
                t0 = time.perf_counter()
                for _ in range(20):
                    e.run_compress()
                e.sync()
                t_local = (time.perf_counter() - t0) / 20 * 1e6
                blk, n = e.export_groups() if mode == "groups" else e.export_block()
                payload.append(np.asarray(blk).reshape(-1)); total += n
            dev = torch.from_numpy(np.stack(payload)).cuda()
            torch.cuda.synchronize()
            merge = (lambda: e.merge_groups(int(dev.data_ptr()), total, n_records=G)) if mode == "groups" else \
                    (lambda: e.merge_gain(int(dev.data_ptr()), total, n_blocks=G))
            for _ in range(3):
                merge(); e.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                merge(); e.sync()
            t_merge = (time.perf_counter() - t0) / 20 * 1e6
            res = e.result()
            out[mode] = (t_local, t_merge, res.dx, res.P_new, dev.numel() * 8 / G)
    d = np.linalg.norm(out["blocks"][2] - out["groups"][2]) / np.linalg.norm(out["blocks"][2])
    print(f"G={G}: blocks: local K1-K5 {out['blocks'][0]:.0f} us, merge+gain {out['blocks'][1]:.0f} us, {out['blocks'][4]/1e3:.0f} KB/rank | "
          f"groups: local {out['groups'][0]:.0f} us, merge+gain {out['groups'][1]:.0f} us, {out['groups'][4]/1e3:.0f} KB/rank | dx diff {d:.1e}", flush=True)
