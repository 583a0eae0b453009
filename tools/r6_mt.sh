#!/bin/bash
# round 6: the macro-tile scoring forward (variant library tmp_var/libmt.so) -- parity at the cfg4 shape, then the cfg4 bench lines with and without it
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-mt}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export G4R_LIB=${LIBV:-$ROOT/tmp_var/libmt.so}
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -m gpu -x -q -n 4 > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log; fi
B="timeout 300 python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-micro"
for c in ${CFGS:-cfg4}; do
echo "== $c: macro tiles + prefetch"; $B --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python tools/benchsum.py $OUT/bench_$c.json
echo "== $c: macro tiles, no prefetch"; G4R_NO_PREFETCH=1 $B --config $c > $OUT/bench_${c}_nopf.json 2> $OUT/bench_${c}_nopf.err; python tools/benchsum.py $OUT/bench_${c}_nopf.json
echo "== $c: 64 x 64 tiles"; G4R_NO_MT=1 $B --config $c > $OUT/bench_${c}_nomt.json 2> $OUT/bench_${c}_nomt.err; python tools/benchsum.py $OUT/bench_${c}_nomt.json
done
