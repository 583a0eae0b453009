#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 50 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/d_${name}_${c}.json 2> $OUT/d_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/d_${name}_${c}.json | tail -1
}
for c in cfg3 cfg4; do
  run base $c 300 G4R_WIDE2=1
  for v in 1 2 4 8 15; do run dbg$v $c 300 G4R_WIDE2=1 G4R_LIB=$ROOT/tmp_var/lib_p1dbg$v.so; done
done
