#!/bin/bash
# round 6: the macro-tile scoring backward (variant library) -- parity at the cfg4 shape, then cfg4 bench lines with and without it
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-bmt}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export G4R_LIB=${LIBV:-$ROOT/tmp_var/libbmt.so}
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -m gpu -x -q -n 4 > $OUT/tests.log 2>&1; tail -15 $OUT/tests.log; fi
B="timeout 300 python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-micro"
for c in ${CFGS:-cfg4}; do
echo "== $c: macro tiles forward + backward"; $B --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python tools/benchsum.py $OUT/bench_$c.json
echo "== $c: backward on 64 x 64 tiles"; G4R_NO_BMT=1 $B --config $c > $OUT/bench_${c}_nobmt.json 2> $OUT/bench_${c}_nobmt.err; python tools/benchsum.py $OUT/bench_${c}_nobmt.json
done
