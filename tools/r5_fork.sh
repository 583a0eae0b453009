#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_wide_layers.py tests/test_gpu_defer.py -x -q 2>&1 | tail -5
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/fk_${name}_${c}.json 2> $OUT/fk_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/fk_${name}_${c}.json
}
run nofork cfg3 1000 G4R_FORK=0
run fork cfg3 1000
run old cfg4 1000 G4R_WIDE2=0
run m24_nofork cfg4 1000 G4R_WIDE2=24 G4R_FORK=0
run m24_fork cfg4 1000 G4R_WIDE2=24
run m8_merged cfg4 1000 G4R_WIDE2=8
run m25_fork cfg4 1000 G4R_WIDE2=25
