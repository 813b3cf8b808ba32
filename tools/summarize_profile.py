#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_*.{csv,md}."""
import csv, glob, json, os, sys
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = f"gpurun_out/profiles_{tag}"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]       # (gpurun merges every run's files into the same directory)
stats = newest(f"{src}/stats/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"{dst}/{tag}_kernel_stats.csv", "w") as f:
    f.write(open(stats).read())
def pmc(kind):
    fs = newest(f"{src}/{kind}/*/*_counter_collection.csv")
    if not fs: return {}
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]; acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}
fetch, write = pmc("fetch"), pmc("write")


def pmc_multi(kind):
    """kernel -> counter -> (mean per launch, launches) for a pass that collected several counters."""
    fs = newest(f"{src}/{kind}/*/*_counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if fs:
        for r in csv.DictReader(open(fs[0])):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: {c: (v[0] / v[1], v[1]) for c, v in d.items()} for k, d in acc.items()}


mfma, sq = pmc_multi("mfma"), pmc_multi("sq")
bench = [l for l in open(f"{src}/stats_bench.log") if l.startswith("{")]
line = json.loads(bench[-1]) if bench else {}
command = open(f"{src}/command.txt").read().strip() if os.path.exists(f"{src}/command.txt") else "python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs"
last_log = [l.strip() for l in open(f"{src}/stats_bench.log") if l.startswith("N=")][-1:]
with open(f"{dst}/{tag}_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary {tag}\n\ncommand: `rocprofv3 --kernel-trace --stats -- {command}` "
            "(+ separate `--pmc` passes: FETCH_SIZE; WRITE_SIZE; matrix-core counters; SQ counters)\n\n")
    if last_log and not line:
        f.write(f"the profiled program's own report: `{last_log[0]}`\n\n")
    f.write("| kernel | calls | avg us | total % | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch |\n|---|---|---|---|---|---|\n")
    for r in rows:
        n = r["Name"].strip('"')
        fe = fetch.get(n, (None,))[0]; wr = write.get(n, (None,))[0]
        f.write(f"| `{n[:70]}` | {r['Calls']} | {float(r['AverageNs'])/1000:.1f} | {float(r['Percentage']):.1f} | "
                f"{'' if fe is None else f'{fe:.0f}'} | {'' if wr is None else f'{wr:.0f}'} |\n")
    f.write("\nFETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); the kernels here read 8 B per lane, "
            "which is uncalibrated, so the raw counter is listed.\n")
    if mfma:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, GRBM_GUI_ACTIVE the busy cycles summed over the 8 XCDs, SQ_BUSY_CYCLES quad-cycles
        # per SE (MI355X_MICROARCH.md); MFMA utilisation of a kernel = matrix-pipe busy cycles / (kernel cycles x 1024 SIMDs).
        f.write("\n## Matrix-core and VALU activity (PMC pass `mfma`, per launch)\n\n")
        f.write("| kernel | MFMA MOPS f64 (x512 flop) | MFMA MOPS f32 | MFMA busy cycles | GRBM_GUI_ACTIVE (sum of 8 XCDs) | waves | "
                "MFMA util % of 1024 SIMDs | f64 MFMA TFLOP/s vs 78.6 peak |\n|---|---|---|---|---|---|---|---|\n")
        dur = {r["Name"].strip('"'): float(r["AverageNs"]) for r in rows}
        for n, d in sorted(mfma.items(), key=lambda kv: -dur.get(kv[0], 0)):
            g = lambda c: d.get(c, (0.0, 0))[0]
            cyc = g("GRBM_GUI_ACTIVE") / 8.0
            util = 100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0) if cyc > 0 else 0.0
            ns = dur.get(n, 0.0)
            tf = g("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512.0 / ns / 1e3 if ns > 0 else 0.0
            f.write(f"| `{n[:60]}` | {g('SQ_INSTS_VALU_MFMA_MOPS_F64'):.0f} | {g('SQ_INSTS_VALU_MFMA_MOPS_F32'):.0f} | "
                    f"{g('SQ_VALU_MFMA_BUSY_CYCLES'):.0f} | {g('GRBM_GUI_ACTIVE'):.0f} | {g('SQ_WAVES'):.0f} | {util:.2f} | {tf:.3f} |\n")
    if sq:
        f.write("\n## Wave time split (PMC pass `sq`, per launch; SQ_* cycle counters are quad-cycles summed over waves)\n\n")
        f.write("| kernel | SQ_WAVE_CYCLES | WAIT_ANY % | WAIT_INST_ANY % | ACTIVE_INST_ANY % | ACTIVE_INST_VALU % | VALU insts | of them FMA_F64 | LDS insts |\n"
                "|---|---|---|---|---|---|---|---|---|\n")
        for n, d in sq.items():
            g = lambda c: d.get(c, (0.0, 0))[0]
            wc = g("SQ_WAVE_CYCLES") or 1.0
            f.write(f"| `{n[:60]}` | {wc:.0f} | {100 * g('SQ_WAIT_ANY') / wc:.1f} | {100 * g('SQ_WAIT_INST_ANY') / wc:.1f} | "
                    f"{100 * g('SQ_ACTIVE_INST_ANY') / wc:.1f} | {100 * g('SQ_ACTIVE_INST_VALU') / wc:.1f} | {g('SQ_INSTS_VALU'):.0f} | "
                    f"{g('SQ_INSTS_VALU_FMA_F64'):.0f} | {g('SQ_INSTS_LDS'):.0f} |\n")
    if line:
        f.write("\nbench line of the profiled run (profiled runs clock lower than un-profiled ones):\n\n```json\n" + json.dumps(line) + "\n```\n")
# machine-readable companion used by bench.py for roofline.traffic
pm = {}
for r in rows:
    n = r["Name"].strip('"')
    pm[n] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1000.0,
             "fetch_kb": fetch.get(n, (None,))[0], "write_kb": write.get(n, (None,))[0]}
    for cname, src_d in (("SQ_INSTS_VALU_FMA_F64", sq), ("SQ_INSTS_VALU", sq), ("SQ_INSTS_VALU_MFMA_MOPS_F64", mfma),
                         ("SQ_INSTS_VALU_MFMA_MOPS_F32", mfma), ("SQ_VALU_MFMA_BUSY_CYCLES", mfma), ("GRBM_GUI_ACTIVE", mfma)):
        v = src_d.get(n, {}).get(cname)
        pm[n][cname] = v[0] if v else None
json.dump({"tag": tag, "kernels": pm, "command": command, "workload": line.get("config", {}).get("workload")}, open(f"{dst}/{tag}_pmc.json", "w"), indent=1)
print(open(f"{dst}/{tag}_summary.md").read())
