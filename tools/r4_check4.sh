#!/bin/bash
# one GPU call: the tests that run wide layers (k_gru_p1<64, ...>), then cfg3 / cfg4 bench lines
mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_widths.py tests/test_gpu_shapes.py tests/test_gpu_eval.py tests/test_gpu_dma_tiles.py -x -q -n 4 2>&1 | tail -5
for c in cfg3 cfg4; do timeout 200 python bench.py --config $c --steps 1500 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r4d/$c.json 2> gpurun_out/r4d/$c.err; echo "== $c"; python tools/benchsum.py gpurun_out/r4d/$c.json; done
