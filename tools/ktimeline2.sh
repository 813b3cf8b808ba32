#!/bin/bash
# usage (on the GPU box): tools/ktimeline2.sh u15|u30|frame|mixed -> kernel timeline of the last resident update of that batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tl2_$1
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/ktimeline_prob.py $1 > $out.log 2>&1
tail -1 $out.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'k_feature' in r['Kernel_Name']]
last = max(i for i in starts if i == 0 or 'k_feature' not in rows[i - 1]['Kernel_Name'])
rows = rows[last:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{r['Kernel_Name'][:66]:66s} {s/1000:9.1f} -> {e/1000:9.1f} us  ({(e-s)/1000:7.1f})  grid {r.get('Grid_Size','?'):>8s} queue {r.get('Queue_Id','?')}")
PY
