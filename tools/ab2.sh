#!/bin/bash
# A/B of variant libraries on the bench lines:  bash tools/ab2.sh "<name>=<path.so> ..." "cfg2 cfg3 cfg4" [extra bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT; cd $ROOT
for rep in 1 2; do
for v in $1; do
  name=${v%%=*}; lib=${v#*=}
  for c in ${2:-cfg2}; do
    steps=3000; [ $c != cfg2 ] && [ $c != cfg1 ] && [ $c != cfg5 ] && steps=1000
    if [ "$lib" = base ]; then unset G4R_LIB; else export G4R_LIB=$ROOT/$lib; fi
    timeout 300 python bench.py --config $c --steps $steps --warmup 300 --no-cpu-baseline --no-micro $3 > $OUT/ab_${name}_${c}_$rep.json 2> $OUT/ab_${name}_${c}_$rep.err
    echo "== $name $c rep $rep"; python tools/benchsum.py $OUT/ab_${name}_${c}_$rep.json
  done
done
done
