#!/bin/bash
# one GPU call: the tests that run the generic / exact-replica sparse update, then the staged bench lines
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_exact_replicas.py tests/test_gpu_parity.py tests/test_gpu_virtual_ranks.py tests/test_gpu_widths.py tests/test_gpu_multirank.py tests/test_gpu_golden.py -x -q -n 4 2>&1 | tail -8
G4R_FORCE_STAGED=1 timeout 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-micro --sparse-exact > gpurun_out/r4b/staged_exact.json 2> gpurun_out/r4b/staged_exact.err; python tools/benchsum.py gpurun_out/r4b/staged_exact.json
