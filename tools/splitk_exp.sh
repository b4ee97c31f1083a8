#!/bin/bash
# experiment: forced split-K of the back-projection at full J: time, then L2-miss traffic for one setting
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/splitk; mkdir -p $OUT
REPO=$PWD
for s in 4 8 16; do
  export PLSHIP_SPLITK=$s
  timeout -k 5 90 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --converge-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('S=$s', round(d['ms_per_step'],3), d['roofline']['per_kernel_ms'])" >> $OUT/times.log
  echo "timed S=$s"
done
export PLSHIP_SPLITK=8
cd /tmp && timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_s8 -o run -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --converge-steps 0 > $OUT/pmc_s8.log 2>&1
echo "pmc done"
