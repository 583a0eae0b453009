#!/bin/bash
# one GPU call: the parity tests that cover every (loss, final activation) pair, then short bench lines of cfg2 / cfg3 / cfg4 / cfg5
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_shapes.py tests/test_gpu_paramfiles.py -x -q -n 4 2>&1 | tail -5
for c in cfg2 cfg3 cfg4 cfg5; do timeout 200 python bench.py --config $c --steps 1500 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r4c/$c.json 2> gpurun_out/r4c/$c.err; echo "== $c"; python tools/benchsum.py gpurun_out/r4c/$c.json; done
