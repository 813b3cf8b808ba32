#!/bin/bash
# Kernel iteration loop ON THE GPU BOX: parity subset first (stop on failure), then stage times of the configs that matter,
# then per-kernel averages of the headline and the north-star size.   usage: tools/kiter.sh [tag] [pytest -k expr]
tag=${1:-it}
kexpr=${2:-"golden or band or ring or ragged or sweep or headline or shipped"}
out=gpurun_out/r3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f32.py -x -q -k "$kexpr" > $out/${tag}_tests.log 2>&1
rc=$?
tail -3 $out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "PARITY FAILED"; exit 1; fi
for cfg in "30 2000 10 50" "30 10000 10 30" "20 500 8 50" "50 20000 15 10"; do
  timeout -k 10 300 python3 tools/one_config.py $cfg 2>&1 | tail -1
done | tee $out/${tag}_stages.log
bash tools/kstats.sh 30 2000 10 30 2>&1 | tail -12 | tee $out/${tag}_kstats_2000.log
bash tools/kstats.sh 30 10000 10 20 2>&1 | tail -12 | tee $out/${tag}_kstats_10000.log
