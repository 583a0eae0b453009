#!/bin/bash
# round 6 iteration call: GPU tests (optional subset) + cfg2 bench line with per-kernel times
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r6}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -m gpu -x -q -n 4 > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log; fi
for c in ${CFGS:-cfg2}; do
timeout 300 python bench.py --config $c --steps 2000 --warmup 200 --no-cpu-baseline --no-micro > $OUT/bench_$c.json 2> $OUT/bench_$c.err
python tools/benchsum.py $OUT/bench_$c.json
done
