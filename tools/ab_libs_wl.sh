#!/bin/bash
# dev tool (GPU box): interleaved A/B of libmopk variants on one bench.py workload: bash tools/ab_libs_wl.sh <workload> <name> [<name> ...]
# ("base" = mop_amd/libmopk.so); two rounds, one process per measurement
WL=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then L=$PWD/mop_amd/libmopk.so; else L=$PWD/mop_amd/libmopk_$v.so; fi
    printf "%-24s " "$v"; MOPK_LIB=$L python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms  core', round(d['roofline']['launch_ms'],3))"
  done
done
