#!/bin/bash
# round-3 measurement pass: virtual-rank rule study, in-kernel phase traces (cfg2 GRU / update, cfg4 scoring), baseline stats
cd "$(dirname "$0")/.."
python tools/virtual_ranks_study.py > gpurun_out/r3_vr_study2.log 2>&1; tail -50 gpurun_out/r3_vr_study2.log
CLK=$PWD/gru4rec_amd/_variants/libgru4rec_hip_clk.so
G4R_LIB=$CLK G4R_CLK=1 python tools/clk.py > gpurun_out/r3_clk_cfg2.txt 2>&1; tail -40 gpurun_out/r3_clk_cfg2.txt
G4R_LIB=$CLK G4R_CLK=1 CFG=cfg4 KERNEL=fwd python tools/clk_score.py > gpurun_out/r3_clk_cfg4_fwd.txt 2>&1; tail -8 gpurun_out/r3_clk_cfg4_fwd.txt
G4R_LIB=$CLK G4R_CLK=1 CFG=cfg4 KERNEL=bwd python tools/clk_score.py > gpurun_out/r3_clk_cfg4_bwd.txt 2>&1; tail -12 gpurun_out/r3_clk_cfg4_bwd.txt
for c in cfg2 cfg3 cfg4; do python bench.py --config $c --steps 1500 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_base_$c.json 2> gpurun_out/r3_base_$c.err; python tools/benchsum.py gpurun_out/r3_base_$c.json; done
