#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of the bench command, then (separately) PMC passes.
# usage: tools/profile_bench.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_ARGS"   # e.g. BENCH_ARGS="--config c3 --ipb-steps 0"
cd /tmp
echo "== kernel trace + stats" | tee "$OUT/log.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH >> "$OUT/log.txt" 2>&1
echo "exit=$?" >> "$OUT/log.txt"
if [ "$2" == "pmc" ]; then
  rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
  i=0
  for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    echo "== pmc pass $i: $CNT" >> "$OUT/log.txt"
    rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pmc$i" -o bench -- $BENCH >> "$OUT/log.txt" 2>&1
    echo "exit=$?" >> "$OUT/log.txt"
  done
fi
find "$OUT" -name "*.csv" | head -50 >> "$OUT/log.txt"
