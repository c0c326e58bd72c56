#!/bin/bash
# dev tool (GPU box): per-gradient errors + kernel-trace stats only (no pytest) -> gpurun_out/$1
set -o pipefail
T=${1:-q}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$T
mkdir -p $OUT
timeout -k 10 300 python tools/check_fused.py > $OUT/check.log 2>&1; grep -A2 "ew_ns_shared_v5_r4_mix5\|ew_odd_shared_v5" $OUT/check.log | cut -c1-400
timeout -k 10 300 python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/tune.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; head -5 $OUT/kernel_stats.csv | cut -c1-160
