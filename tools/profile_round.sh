#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for HBM bytes.
# usage: tools/profile_round.sh r01   -> gpurun_out/profiles_r01/{stats,fetch,write}/...
tag=${1:-r01}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $CMD > $out/fetch_bench.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $CMD > $out/write_bench.log 2>&1
ls -R $out | head -40
