#!/bin/bash
# usage: tools/profile_cmd.sh <tag> <counters...> -- python3 script args   (runs kernel-trace+stats, then one PMC pass)
TAG=$1; shift
CNT=()
while [ "$1" != "--" ]; do CNT+=("$1"); shift; done
shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD=("$@"); CMD[1]="$PWD/${CMD[1]}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- "${CMD[@]}" > "$OUT/log.txt" 2>&1
rocprofv3 --pmc "${CNT[@]}" --output-format csv -d "$OUT/pmc" -o run -- "${CMD[@]}" >> "$OUT/log.txt" 2>&1
echo done >> "$OUT/log.txt"
