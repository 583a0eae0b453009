#!/bin/bash
# round 6, call 1: launch-chain probe + baseline stage trace of the cfg2 GRU pair
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r6a
mkdir -p $OUT
cd $ROOT
timeout 120 tools/probes/chain_probe > $OUT/chain_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/out_p
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_p -- $ROOT/tools/probes/chain_probe > $OUT/chain_probe_rocprof.log 2>&1
cp /tmp/out_p/*/*kernel_stats.csv $OUT/chain_probe_kernel_stats.csv 2>/dev/null
cd $ROOT
G4R_LIB=$ROOT/tmp_var/libclk.so G4R_CLK=1 timeout 200 python tools/clk.py > $OUT/clk_cfg2_baseline.txt 2>&1
timeout 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-micro > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python tools/benchsum.py $OUT/bench_cfg2.json
