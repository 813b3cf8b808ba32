#!/usr/bin/env python3
"""Randomised soak of the sharded update on ONE GPU with logical shards and RAGGED tracks (tools/soak_holes.py's generator):
the root-block exchange (msckf_run_compress + msckf_export_block on every shard, msckf_run_merge_gain) for any batch, and the
shipped group-record exchange (tests/test_gpu_parity.py::_shipped_merge) where every track fits 15 clone slots.
usage: soak_shards.py [cases] [seed]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import msckf_amd  # noqa: F401
from msckf_amd.api import UpdateEngine
from msckf_amd.shard import partition_features
from oracle import msckf_oracle as oracle
sys.path.insert(0, os.path.join(ROOT, "tools")); from soak_holes import ragged


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    import test_gpu_parity as tp
    bad = 0
    with UpdateEngine(max_clones=53, max_features=2048, max_track=31) as eng:
        for c in range(cases):
            N = int(rng.integers(3, 54)); F = int(rng.integers(8, 300)); S = int(rng.choice([2, 3, 4, 8]))
            hi = int(rng.integers(2, min(N, 31) + 1))
            prob = ragged(rng, N, F, 2, hi, float(rng.choice([0.0, 0.1, 0.4])))
            ref = oracle.update(prob, dense_noise=False)
            shards = partition_features(prob.view_ptr, S)
            # root blocks
            try:
                blocks, total, acc = [], 0, np.zeros(prob.F, dtype=np.uint8)
                for lo, hi_ in shards:
                    eng.load(prob.subset(lo, hi_))
                    eng.run_compress()
                    blk, n = eng.export_block()
                    acc[lo:hi_] = eng.result().accepted
                    blocks.append(blk); total += n
                eng.set_state(prob)
                eng.merge_gain(np.stack(blocks), total)
                res = eng.result()
                ok = res.status == ref["status"] and np.array_equal(acc, ref["accepted"])
                e = max(rel(res.dx, ref["dx"]), rel(res.P_new, ref["P_new"])) if ok and res.status == 0 else 0.0
                if not ok or e > 1e-8:
                    bad += 1
                    print(f"case {c} blocks: N={N} F={F} S={S} views<={hi}: status {res.status}/{ref['status']} mask {np.array_equal(acc, ref['accepted'])} err {e:.2e}", flush=True)
            except Exception as ex:
                bad += 1
                print(f"case {c} blocks: N={N} F={F} S={S} views<={hi}: {type(ex).__name__} {ex}", flush=True)
            # group records (every track within 15 slots)
            if eng.band_ok(prob):
                try:
                    tp._shipped_merge(eng, prob, shards, ref)
                except Exception as ex:
                    bad += 1
                    print(f"case {c} groups: N={N} F={F} S={S} views<={hi}: {type(ex).__name__} {str(ex)[:200]}", flush=True)
                    eng.set_exchange_mask(None); eng.set_exchange_span(0); eng.set_group_exchange(False)
    print(f"{cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
