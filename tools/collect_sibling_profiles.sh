#!/bin/bash
# Run on the GPU box (gpurun): bench line + rocprofv3 kernel-trace stats of `bench.py --workload quartet|whisper` (BASELINE configs[3] / [4])
# -> gpurun_out/$1_{quartet,whisper}/ ; afterwards `bash tools/collect_sibling_profiles.sh --stage $1` copies the summaries into profiles/.
set -o pipefail
if [ "$1" = "--stage" ]; then
  R=${2:-r03}
  for wl in quartet whisper; do
    cp gpurun_out/${R}_$wl/kernel_stats.csv profiles/${R}_${wl}_kernel_stats.csv
    tail -1 gpurun_out/${R}_$wl/bench.json > profiles/${R}_${wl}_bench.json
  done
  exit 0
fi
R=${1:-r03}
export TMPDIR=/tmp
for wl in quartet whisper; do
  OUT=$PWD/gpurun_out/${R}_$wl
  mkdir -p $OUT
  python3 bench.py --workload $wl --steps 2 --warmup 2 --no-cpu-baseline > $OUT/tune.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1 || exit 1
  cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv
  python3 bench.py --workload $wl --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || exit 1
  tail -1 $OUT/bench.json | cut -c1-400
done
