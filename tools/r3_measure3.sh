#!/bin/bash
cd "$(dirname "$0")/.."
timeout 150 python tools/sk_debug.py > gpurun_out/r3_sk_debug.txt 2>&1; tail -6 gpurun_out/r3_sk_debug.txt
timeout 400 python -m pytest tests/test_gpu_dma_tiles.py tests/test_gpu_baseline_configs.py tests/test_gpu_catalogue_scale.py "tests/test_gpu_parity.py::test_many_negatives_big_batch" -q -m gpu -x --timeout 150 -p no:cacheprovider > gpurun_out/r3_t5.log 2>&1; tail -5 gpurun_out/r3_t5.log
for v in "G4R_STREAMK=0 G4R_PF_BWD=0" "G4R_STREAMK=1 G4R_PF_BWD=0" "G4R_STREAMK=1 G4R_PF_BWD=1"; do for c in cfg4 cfg3; do env $v timeout 150 python bench.py --config $c --steps 1000 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; echo "== $v $c"; python tools/benchsum.py gpurun_out/r3_ab.json; done; done
