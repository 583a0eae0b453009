#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_p2p.py -q -m gpu -x --timeout 200 -p no:cacheprovider -s > gpurun_out/r3_p2p_tests.log 2>&1; tail -25 gpurun_out/r3_p2p_tests.log
echo "== clk bwd2 cfg4"
G4R_LIB=gru4rec_amd/_variants/libgru4rec_hip_clk.so G4R_CLK=1 CFG=cfg4 KERNEL=bwd timeout 100 python tools/clk_score.py 2>&1 | tail -12
