#!/bin/bash
# usage (on the GPU box): tools/ktimeline.sh N F M [dtype]  -> start / end of every kernel of the LAST resident update of one
# config, relative to its first kernel (rocprofv3 --kernel-trace): shows what runs beside what (root sweep | k_gain_stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tl_$1_$2_$3
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/one_config.py $1 $2 $3 3 $4 > $out.log 2>&1
tail -1 $out.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last update = the last run of kernels that starts with k_feature
starts = [i for i, r in enumerate(rows) if 'k_feature' in r['Kernel_Name']]
rows = rows[starts[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{r['Kernel_Name'][:70]:70s} {s/1000:9.1f} -> {e/1000:9.1f} us  ({(e-s)/1000:7.1f})  queue {r.get('Queue_Id','?')}")
PY
