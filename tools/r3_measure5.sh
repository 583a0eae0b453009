#!/bin/bash
cd "$(dirname "$0")/.."
CLK=$PWD/gru4rec_amd/_variants/libgru4rec_hip_clk.so
for nst in 3 4; do echo "== stream-K NST=$nst"; G4R_SK_NST=$nst G4R_LIB=$CLK G4R_CLK=1 CFG=cfg4 KERNEL=fwd timeout 120 python tools/clk_score.py > gpurun_out/r3_clk_sk$nst.txt 2>&1; tail -9 gpurun_out/r3_clk_sk$nst.txt; done
