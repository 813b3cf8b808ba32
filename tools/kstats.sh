#!/bin/bash
# usage (on the GPU box): tools/kstats.sh N F M iters [dtype]  -> per-kernel averages of one config (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3/ks_$1_$2_$3
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/one_config.py $1 $2 $3 $4 $5 > $out.log 2>&1
tail -1 $out.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
