#!/bin/bash
# Round 5, wide-layer kernels: parity tests, then A/B of the round-1 kernels (G4R_WIDE2=0) against the split-K ones on cfg3 / cfg4,
# then slice-length variants.  bash tools/r5_wide.sh [notest]
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
if [ "$1" != notest ]; then
  timeout 900 python -m pytest tests/test_gpu_wide_layers.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -25 | tee $OUT/wide_tests.log
fi
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 200 --no-cpu-baseline --no-micro > $OUT/ab_${name}_${c}.json 2> $OUT/ab_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/ab_${name}_${c}.json
}
for c in cfg3 cfg4; do
  run old $c 1000 G4R_WIDE2=0
  run new $c 1000 G4R_WIDE2=31
done
run p1ks128 cfg3 600 G4R_P1_KS=128
run p1ks512 cfg3 600 G4R_P1_KS=512
run bbks192 cfg3 600 G4R_BB_KS=192
run bbks384 cfg3 600 G4R_BB_KS=384
run p2ks64 cfg3 600 G4R_P2_KS=64 G4R_BA_KS=64
run p2ks256 cfg3 600 G4R_P2_KS=256 G4R_BA_KS=256
run p1ks256 cfg4 600 G4R_P1_KS=256
run bbks256 cfg4 600 G4R_BB_KS=256
run p2ks128 cfg4 600 G4R_P2_KS=128 G4R_BA_KS=128
