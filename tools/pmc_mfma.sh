#!/bin/bash
# MFMA / LDS utilisation counters of the step kernels:  bash tools/pmc_mfma.sh <cfg> <tag> [env...]
CFG=${1:-cfg4}; TAG=${2:-base}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config $CFG --steps 60 --warmup 20 --no-cpu-baseline --no-micro --profile-steps 0 --no-graph"
rm -rf /tmp/p1 /tmp/p2
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/p1 -- $B > $OUT/p1.log 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/p2 -- $B > $OUT/p2.log 2>&1
python $ROOT/tools/pmc_counters.py $OUT/counters_${CFG}.json /tmp/p1/*/*counter_collection.csv /tmp/p2/*/*counter_collection.csv
