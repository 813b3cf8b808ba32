#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes (HBM bytes; matrix-core /
# VALU activity and wave occupancy).  --pmc passes never carry trace flags (gpurun refuses the combination).
# usage: tools/profile_round.sh <tag> [N F M iters dtype]
#   no workload: the headline through bench.py            -> gpurun_out/profiles_<tag>/{stats,fetch,write,mfma,sq}/...
#   with one   : that config through tools/one_config.py  (e.g. r04_ns 30 10000 10 30 f64; r04_cfg4 50 20000 15 10 f64)
# then, in the build container: python3 tools/summarize_profile.py <tag>
tag=${1:-r04}
shift
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ $# -ge 3 ]; then
  CMD="python3 tools/one_config.py $*"
else
  CMD="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs"
fi
echo "$CMD" > $out/command.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats_bench.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $CMD > $out/fetch_bench.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $CMD > $out/write_bench.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- $CMD > $out/mfma_bench.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- $CMD > $out/sq_bench.log 2>&1
tail -1 $out/stats_bench.log
ls $out
