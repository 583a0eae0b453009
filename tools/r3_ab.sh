#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_catalogue_scale.py tests/test_gpu_multirank.py tests/test_gpu_widths.py tests/test_gpu_shapes.py tests/test_gpu_e2e_recall.py -q -m gpu -x --timeout 150 -p no:cacheprovider > gpurun_out/r3_t9.log 2>&1; tail -3 gpurun_out/r3_t9.log
for c in cfg2 cfg5 cfg3 cfg4; do echo "== $c"; timeout 120 python bench.py --config $c --steps 3000 --warmup 300 --no-cpu-baseline --no-micro --long-steps 0 > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json; done
