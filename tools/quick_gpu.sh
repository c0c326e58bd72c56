#!/bin/bash
# dev tool (GPU box): edgewise parity tests + per-gradient errors + short bench + kernel-trace stats -> gpurun_out/$1
set -o pipefail
T=${1:-q}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$T
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_edgewise.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
timeout -k 10 300 python tools/check_fused.py > $OUT/check.log 2>&1; tail -12 $OUT/check.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-900
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; head -12 $OUT/kernel_stats.csv | cut -c1-200
