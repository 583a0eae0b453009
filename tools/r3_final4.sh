#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/final4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg4 cfg3; do
  rm -rf /tmp/out_s
  timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- python $ROOT/bench.py --config $c --steps 600 --warmup 100 --no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0 > $OUT/stats_$c.log 2>&1
  cp /tmp/out_s/*/*kernel_stats.csv $OUT/r03_kernel_stats_rocprofv3_$c.csv 2>/dev/null; echo "== $c"; head -6 $OUT/r03_kernel_stats_rocprofv3_$c.csv | cut -c1-150
done
