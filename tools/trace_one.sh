#!/bin/bash
# usage: tools/trace_one.sh <outdir>  -- rocprofv3 kernel trace of a short bench run + per-launch table of one update
out=${1:-gpurun_out/prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/bench.log 2>&1
python3 - $out <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
idx=[i for i,r in enumerate(rows) if 'k_feature' in r['Kernel_Name']][-3]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:idx+17]:
    if 'k_feature' in r['Kernel_Name'] and int(r['Start_Timestamp'])>t0: break
    print(f"{r['Kernel_Name'][:60]:60s} start={(int(r['Start_Timestamp'])-t0)/1000:8.1f}us dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:7.1f}us grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
PY
