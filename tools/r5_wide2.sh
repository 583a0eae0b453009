#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
timeout 120 python tools/r5_mean_debug.py 2>&1 | tail -90 > $OUT/mean_debug.txt; grep -c BAD $OUT/mean_debug.txt
timeout 600 python -m pytest tests/test_gpu_wide_layers.py -x -q 2>&1 | tail -8
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/w2_${name}_${c}.json 2> $OUT/w2_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/w2_${name}_${c}.json
}
for c in cfg3 cfg4; do
  run old $c 600 G4R_WIDE2=0
  run m16 $c 600 G4R_WIDE2=16
  run m24 $c 600 G4R_WIDE2=24
  run m24ks128 $c 600 G4R_WIDE2=24 G4R_BB_KS=128
  run m24ks256 $c 600 G4R_WIDE2=24 G4R_BB_KS=256
done
