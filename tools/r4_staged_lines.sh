#!/bin/bash
# the three N > 1-data-path bench lines on a one-rank communicator (what tools/final_profile.sh also runs) -> gpurun_out/final/
OUT=gpurun_out/final; mkdir -p $OUT; B="python bench.py"
G4R_FORCE_STAGED=1 timeout 200 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > $OUT/r04_bench_staged_1rank.json 2> $OUT/bench_staged.err; echo "== staged, gpu-local rows"; python tools/benchsum.py $OUT/r04_bench_staged_1rank.json
G4R_FORCE_STAGED=1 timeout 200 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro --sparse-exact > $OUT/r04_bench_staged_1rank_exact.json 2> $OUT/bench_staged_exact.err; echo "== staged, exact replicas"; python tools/benchsum.py $OUT/r04_bench_staged_1rank_exact.json
G4R_FORCE_STAGED=1 timeout 240 $B --config cfg4 --steps 600 --warmup 100 --no-cpu-baseline --no-micro > $OUT/r04_bench_staged_1rank_cfg4.json 2> $OUT/bench_staged_cfg4.err; echo "== staged cfg4"; python tools/benchsum.py $OUT/r04_bench_staged_1rank_cfg4.json
G4R_FORCE_STAGED=1 timeout 100 $B --steps 20 --warmup 5 --no-cpu-baseline --no-micro > $OUT/r04_bench_staged_1rank_driver_shape.json 2> $OUT/bench_staged_ds.err; echo "== staged, driver shape"; python tools/benchsum.py $OUT/r04_bench_staged_1rank_driver_shape.json
tail -2 $OUT/bench_staged*.err
