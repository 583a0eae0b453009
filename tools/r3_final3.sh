#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final3; mkdir -p $OUT
T0=$(date +%s)
timeout 200 python -m pytest tests -q -m gpu -n 4 --dist loadfile --timeout 180 -p no:cacheprovider > $OUT/tests.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -5 $OUT/tests.log
for c in cfg4 cfg3; do
  timeout 60 python bench.py --config $c --steps 1500 --warmup 200 --no-cpu-baseline --long-steps 0 > $OUT/r03_bench_$c.json 2> $OUT/bench_$c.err
  echo "== $c"; python tools/benchsum.py $OUT/r03_bench_$c.json
done
echo "elapsed $(( $(date +%s) - T0 )) s"
