#!/usr/bin/env python3
"""bench.py -- MSCKF measurement-updates/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one complete measurement update (K1..K7: per-feature Jacobian stack,
nullspace projection, chi-square gate, QR compression, gain, Joseph covariance
update) of ONE filter over one synthetic feature batch, inputs resident in HBM.
N = 1 runs BASELINE.json configs[2] (headline: N=30 clones, 2000 features, track
10, fp64).  N > 1 (launched by torch.distributed.run, one rank per GPU) runs the
feature-sharded update: 2000 features per rank (configs[3] at 4 GPUs), one RCCL
gather of the compressed R blocks to rank 0, serial gain on rank 0, broadcast of
dx / P+; `value` counts 2000-feature update equivalents (weak scaling).

Rank 0 prints ONE JSON line.  The oracle (oracle/msckf_oracle.py) is only timed
as the CPU baseline; it is never on the measured GPU path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix peak (AMD public; half the 157.3 TF FP32 rate)


def algorithmic_costs(N, F, M):
    """Canonical per-update bytes / flops of the reference's dense formulation,
    SURVEY.md section 8(d) (all features accepted)."""
    s = 8
    d, dc, q = 15 + 6 * N, 6 * N, 2 * M - 3
    m = F * q
    bytes_inputs = s * (F * (2 * M + 7) + 24 * N + 2 * d * d + d) + 4 * F * M + F
    bytes_stack = F * q * (d + 1) * s                       # written once (K4), read once (K5)
    fl_A = F * (200 * M + 36 * M + 24 * M * (6 * M + 1) + 2 * q * (6 * M) ** 2 + 12 * q * q * M + q ** 3 / 3 + 2 * q * q)
    fl_B = 2 * m * dc * dc - (2.0 / 3.0) * dc ** 3 + 4 * m * dc
    fl_C = 6 * dc * d * d + 4 * dc * dc * d + (2.0 / 3.0) * dc ** 3 + 4 * d ** 3
    t_roof = (bytes_inputs + bytes_stack) / (HBM_PEAK_GBS * 1e9) + fl_B / (FP64_PEAK_TFLOPS * 1e12) \
        + fl_C / (FP64_PEAK_TFLOPS * 1e12)
    return dict(bytes=bytes_inputs + 2 * bytes_stack, flops_A=fl_A, flops_B=fl_B, flops_C=fl_C,
                bytes_A=bytes_inputs + bytes_stack, t_roof_s=t_roof, rows=m)


def pmc_traffic(kernel_prefixes):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command; raw counters, see the note in profiles/*_summary.md).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    tot, calls = 0.0, 0
    for name, k in d["kernels"].items():
        if any(pfx in name for pfx in kernel_prefixes) and k["fetch_kb"] is not None and k["write_kb"] is not None:
            tot += (k["fetch_kb"] + k["write_kb"]) * 1024.0 * k["calls"]
            calls += k["calls"]
    return tot / calls if calls else None


def cpu_baseline(prob, reps=10):
    """The oracle (NumPy restatement of the reference path) on this box's host
    cores: reference-faithful per-feature Python loop, SVD nullspace, np.linalg.qr,
    explicit inverses, Joseph form -- minus the dense sigma^2*eye(m) allocation
    (9.2 GB at the headline; R_n = sigma^2 I analytically, BASELINE.md section 3 mode ii)."""
    from oracle import msckf_oracle as oracle
    np.linalg.qr(np.random.default_rng(0).standard_normal((400, 60)))      # LAPACK warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = oracle.update(prob, dense_noise=False)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    try:                                                    # threads the BLAS / LAPACK calls of the oracle may use
        from threadpoolctl import threadpool_info
        cores = max([int(p.get("num_threads", 1)) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    return out, dict(value=1.0 / t, unit="updates/s", cores=cores, kind="port",
                     sample=f"{reps} full updates of the same workload (median {t:.2f} s each, {sum(ts):.0f} s in all); "
                            "per-feature stage is a single-threaded Python loop, QR / products use the BLAS "
                            f"thread pool ({cores} threads, {os.cpu_count()} logical CPUs); oracle with "
                            "R_n = sigma^2 I analytic instead of the reference's dense sigma^2*eye(m)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--clones", type=int, default=30)
    ap.add_argument("--features", type=int, default=2000, help="features per GPU")
    ap.add_argument("--track", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded code path even at world size 1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    N, Fg, M = args.clones, args.features, args.track

    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch                                           # plumbing: rendezvous + RCCL gather
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import msckf_amd  # noqa: F401
    from msckf_amd import synth
    from msckf_amd.api import UpdateEngine

    prob = synth.make_problem(N, Fg * world, M, seed=0)
    eng = UpdateEngine(max_clones=N, max_features=Fg * max(world, 1), max_track=max(M, 2), device=local_rank)
    costs = algorithmic_costs(N, Fg, M)

    if not use_dist:
        eng.load(prob)                                           # inputs resident in HBM before the timed region
        for _ in range(args.warmup):
            eng.run()
        eng.sync()
        t0 = time.perf_counter()
        ms_ev, _ = eng.run_timed(args.steps)                     # K steps, HIP events on the engine's stream
        eng.sync()
        wall = time.perf_counter() - t0
        _, stages = eng.run_timed(min(args.steps, 50), stages=True)
        res = eng.result()
        # host-inclusive rate: host arrays in -> dx, P+, mask on host (the drop-in call)
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            one = eng.update_problem(prob)
        host_inclusive = reps / (time.perf_counter() - t1)
        units = args.steps
        seconds = wall
        stats = one.stats
    else:
        from msckf_amd.shard import partition_features
        lo, hi = partition_features(prob.view_ptr, world)[rank]
        local = prob.subset(lo, hi)
        # exchange format (every rank sees the whole batch, so all agree): group triangles when the batch
        # runs the band pipeline -- each rank stops in front of its root sweep, rank 0 folds the shards' groups
        # and runs ONE root sweep -- else the root blocks [R | Q^T r] and the fold-tree merge
        groups = eng.band_ok(prob)
        eng.set_group_exchange(groups)
        eng.load(local)
        nblk = eng.group_record_doubles() if groups else eng.block_doubles()
        d = prob.d
        # send buffer: a group record carries the shard's accepted count itself; a root block [R | Q^T r] gets one
        # trailing double for it (so the gather stays the only collective before the merge)
        rec = nblk if groups else nblk + 1
        mine = torch.zeros(rec, dtype=torch.float64, device="cuda")
        gathered = torch.zeros(world * rec, dtype=torch.float64, device="cuda") if rank == 0 else None
        glist = list(gathered.view(world, rec).unbind(0)) if rank == 0 else None
        packed = torch.zeros(world * nblk, dtype=torch.float64, device="cuda") if rank == 0 and world > 1 and not groups else None
        # broadcast buffer: on rank 0 the engine's own result range dx | P+ (zero copy), a plain tensor elsewhere
        out = torch.as_tensor(eng.result_device_view(), device="cuda") if rank == 0 else \
            torch.zeros(d + d * d, dtype=torch.float64, device="cuda")

        def step():
            eng.run_compress()                                   # K1-K5 on the local shard (no root sweep with groups)
            if groups:
                eng.export_groups(dst_ptr=mine.data_ptr(), count=False)   # D2D into the torch-owned send buffer (syncs)
            else:
                _, n = eng.export_block(dst_ptr=mine.data_ptr())
                mine[nblk] = float(n)
            dist.gather(mine, gather_list=glist, dst=0)          # ONE RCCL gather
            if rank == 0:
                if groups:
                    torch.cuda.current_stream().synchronize()    # the gather has landed (the engine has its own stream)
                    eng.merge_groups(int(gathered.data_ptr()), -1, n_records=world)   # counts are in the records
                else:
                    g2 = gathered.view(world, rec)
                    total = int(g2[:, nblk].sum().item())        # syncs: the gather has landed
                    if world > 1:
                        packed.view(world, nblk).copy_(g2[:, :nblk])   # contiguous blocks for the merge
                        torch.cuda.current_stream().synchronize()
                        src = packed
                    else:
                        src = gathered
                    eng.merge_gain(int(src.data_ptr()), total, n_blocks=world)
                eng.sync()                                       # dx | P+ are in the engine's result range = `out`
            dist.broadcast(out, src=0)                           # state for the next update on every rank

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        wall = time.perf_counter() - t0
        tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        seconds = float(tmax.item())
        units = args.steps * world                               # 2000-feature update equivalents
        stages = None
        host_inclusive = None
        ms_ev = seconds * 1e3
        stats = {}

    if rank == 0:
        line = {
            "metric": "MSCKF measurement-updates/sec (N=30 clones, 2000 features, track=10)",
            "value": units / seconds,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * seconds / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"N={N} clones, F={Fg} features per GPU, track={M}, fp64"
                                   + ("" if world == 1 else f", feature-sharded over {world} GPUs "
                                      f"({Fg * world} features per update), 1 RCCL gather + broadcast per update"),
                       **({"exchange": "group triangles" if groups else "root blocks"} if use_dist else {}),
                       "unit_definition": "one 2000-feature measurement update (K1-K7)", "seed": 0},
        }
        if not use_dist:
            us_step = 1e6 * seconds / args.steps
            us_qr = stages[1]
            n_lv = max(1, stats.get("n_levels", 1))
            line["roofline"] = {
                "kernel": "K5 QR compression: k_fold leaves + k_sweep group merges and root sweep "
                          "(%d launches per update)" % n_lv,
                "bound": "mfma", "unit": "TFLOP/s",
                "achieved": costs["flops_B"] / (us_qr * 1e-6) / 1e12,
                "peak": FP64_PEAK_TFLOPS,
                "frac": costs["flops_B"] / (us_qr * 1e-6) / 1e12 / FP64_PEAK_TFLOPS,
                "traffic": pmc_traffic(("k_fold<", "k_sweep<")),
                "flops_per_launch": costs["flops_B"] / n_lv,
                "avg_launch_us": us_qr / n_lv,
            }
            line["pipeline_roofline"] = {"t_roof_us": costs["t_roof_s"] * 1e6, "t_measured_us": us_step,
                                         "frac": costs["t_roof_s"] * 1e6 / us_step,
                                         "hbm_gbs_algorithmic": costs["bytes"] / (us_step * 1e-6) / 1e9}
            line["stages_us"] = {"feature_K1_K4": stages[0], "qr_K5": stages[1], "gain_K6_K7": stages[2],
                                 "hip_event_ms_per_step": ms_ev / args.steps}
            line["host_inclusive_updates_per_s"] = host_inclusive
            line["accepted"] = int(res.accepted.sum())
            line["plan"] = {"leaves": stats.get("n_leaves"), "levels": stats.get("n_levels"),
                            "host_prep_us": stats.get("us_host_prep"), "h2d_us": stats.get("us_h2d")}
            # f1, reported beside the headline, never inside `value`: the selection +
            # triangulation kernel on the same tracks, and the fused select -> update pass.
            tracks = synth.make_tracks(prob, 0, lost_fraction=0.5)
            sp = synth.SelectParams(min_parallax_deg=8.0)
            eng.load(prob)
            eng.set_tracks(tracks)
            eng.run_select(sp, prob.K)
            us_sel = eng.time_select(50)
            eng.run()
            eng.sync()
            ms_masked, _ = eng.run_timed(50)                       # K1-K7 over the valid subset, HIP events
            us_fused = us_sel + ms_masked / 50 * 1e3
            t2 = time.perf_counter()
            eng.replan()                                           # tree over the valid features only (syncs)
            us_replan = (time.perf_counter() - t2) * 1e6
            eng.run()
            eng.sync()
            ms_replanned, _ = eng.run_timed(50)
            n_views = int(prob.view_ptr[-1])
            sel_bytes = n_views * (7 * 8 + 4) + prob.F * (3 * 4 + 1 + 7 * 8)
            line["select_f1"] = {"kernel": "k_select (get_valid_features)", "us_per_launch": us_sel,
                                 "candidates": prob.F, "valid": int(eng.selection().valid.sum()),
                                 "bytes_algorithmic": sel_bytes, "hbm_gbs_algorithmic": sel_bytes / (us_sel * 1e-6) / 1e9,
                                 "fused_select_update_us": us_fused,
                                 "replan_host_us": us_replan,
                                 "fused_replanned_us": us_sel + us_replan + ms_replanned / 50 * 1e3}
            # f2 / f3: the covariance steps either side of the update on the resident P (host clock
            # around async launches + one sync; augment / remove include their pose upload and sync)
            rng = np.random.default_rng(0)
            eng.set_prior(prob.P, prob.gravity, prob.K, prob.sigma, prob.cam_R, prob.cam_t)
            Phi = np.eye(15) + 1e-3 * rng.standard_normal((15, 15))
            Qd = 1e-8 * np.eye(15)
            for _ in range(10):
                eng.propagate(Phi, Qd)
            eng.sync()
            t3 = time.perf_counter()
            for _ in range(200):
                eng.propagate(Phi, Qd)
            eng.sync()
            us_prop = (time.perf_counter() - t3) / 200 * 1e6
            J15 = np.zeros((6, 15)); J15[:3, :3] = np.eye(3); J15[3:, 12:] = np.eye(3)
            t4 = time.perf_counter()
            for _ in range(20):
                eng.remove_clones([0])
                eng.augment(J15, prob.cam_R[0], prob.cam_t[0])
            us_window = (time.perf_counter() - t4) / 20 * 1e6
            line["resident_f2_f3"] = {"propagate_us": us_prop, "remove_plus_augment_us": us_window,
                                      "clones": N, "bytes_per_propagate": (2 * 15 * prob.d * 2 + prob.d * prob.d * 2) * 8}
            # the CPU baseline runs last: its BLAS threads keep spinning and would disturb host-clocked numbers
            if not args.no_cpu_baseline:
                ref, cpu = cpu_baseline(prob)
                line["cpu_baseline"] = cpu
                e_dx = float(np.linalg.norm(res.dx - ref["dx"]) / np.linalg.norm(ref["dx"]))
                e_P = float(np.linalg.norm(res.P_new - ref["P_new"]) / np.linalg.norm(ref["P_new"]))
                line["parity_vs_cpu_baseline"] = {"dx_rel": e_dx, "P_rel": e_P}
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
