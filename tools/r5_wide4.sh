#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_wide_layers.py -x -q 2>&1 | tail -5
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/w4_${name}_${c}.json 2> $OUT/w4_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/w4_${name}_${c}.json
}
for c in cfg3 cfg4; do
  run auto $c 600
  run auto_ks64 $c 600 G4R_P1_KS=64
  run auto_ks96 $c 600 G4R_P1_KS=96
done
