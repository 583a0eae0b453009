#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for c in cfg4 cfg2 cfg3; do
  rm -rf /tmp/p1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/p1 -- python $ROOT/bench.py --config $c --steps 60 --warmup 20 --no-cpu-baseline --no-micro --profile-steps 0 --no-graph --long-steps 0 > $ROOT/gpurun_out/r3_pmc_$c.log 2>&1
  python $ROOT/tools/pmc_mfma_summary.py $ROOT/gpurun_out/r03_pmc_mfma_$c.json /tmp/p1 $c
done
cd $ROOT
echo "== cfg4s"; timeout 150 python bench.py --config cfg4s --steps 1000 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json
