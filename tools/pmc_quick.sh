#!/bin/bash
# dev tool (GPU box): FETCH / WRITE / SQ counter passes of bench.py (each its own run) -> gpurun_out/$1/pmc_*.txt
set -o pipefail
T=${1:-p}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$T
mkdir -p $OUT
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/tune.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq1.log 2>&1 || exit 1
for p in pmc_fetch pmc_write pmc_sq1; do python3 tools/pmc_summary.py "$OUT/$p/*/*counter_collection.csv" > $OUT/$p.txt; done
grep -A1 "ew_fused" $OUT/pmc_fetch.txt $OUT/pmc_write.txt | grep -v "^--"
