#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_wide_layers.py tests/test_gpu_exact_replicas.py::test_mean_form_against_the_oracle_run_as_replicas tests/test_gpu_exact_replicas.py::test_reduce_form_against_the_oracle_run_as_replicas tests/test_gpu_exact_replicas.py::test_reduce_form_poisons_the_cost_when_the_ranks_negatives_differ tests/test_gpu_mutation.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -12
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/w3_${name}_${c}.json 2> $OUT/w3_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/w3_${name}_${c}.json
}
for c in cfg3 cfg4; do
  run old $c 600 G4R_WIDE2=0
  run auto $c 600
  run auto_ks256 $c 600 G4R_P1_KS=256
  run auto_ks64 $c 600 G4R_P1_KS=64
done
run m9 cfg4 600 G4R_WIDE2=9
run m25 cfg4 600 G4R_WIDE2=25
