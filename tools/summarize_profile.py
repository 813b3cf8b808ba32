#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_*.{csv,md}."""
import csv, glob, json, os, sys
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/profiles_{tag}"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
stats = glob.glob(f"{src}/stats/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"{dst}/{tag}_kernel_stats.csv", "w") as f:
    f.write(open(stats).read())
def pmc(kind):
    fs = glob.glob(f"{src}/{kind}/*/*_counter_collection.csv")
    if not fs: return {}
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]; acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}
fetch, write = pmc("fetch"), pmc("write")
bench = [l for l in open(f"{src}/stats_bench.log") if l.startswith("{")]
line = json.loads(bench[-1]) if bench else {}
with open(f"{dst}/{tag}_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary {tag}\n\ncommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline` "
            "(+ separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)\n\n")
    f.write("| kernel | calls | avg us | total % | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch |\n|---|---|---|---|---|---|\n")
    for r in rows:
        n = r["Name"].strip('"')
        fe = fetch.get(n, (None,))[0]; wr = write.get(n, (None,))[0]
        f.write(f"| `{n[:70]}` | {r['Calls']} | {float(r['AverageNs'])/1000:.1f} | {float(r['Percentage']):.1f} | "
                f"{'' if fe is None else f'{fe:.0f}'} | {'' if wr is None else f'{wr:.0f}'} |\n")
    f.write("\nFETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); the kernels here read 8 B per lane, "
            "which is uncalibrated, so the raw counter is listed.\n")
    if line:
        f.write("\nbench line of the profiled run (profiled runs clock lower than un-profiled ones):\n\n```json\n" + json.dumps(line) + "\n```\n")
# machine-readable companion used by bench.py for roofline.traffic
pm = {}
for r in rows:
    n = r["Name"].strip('"')
    pm[n] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1000.0,
             "fetch_kb": fetch.get(n, (None,))[0], "write_kb": write.get(n, (None,))[0]}
json.dump({"tag": tag, "kernels": pm, "workload": line.get("config", {}).get("workload")}, open(f"{dst}/{tag}_pmc.json", "w"), indent=1)
print(open(f"{dst}/{tag}_summary.md").read())
