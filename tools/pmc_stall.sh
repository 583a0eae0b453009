#!/bin/bash
# stall / issue counters of the step kernels:  bash tools/pmc_stall.sh <cfg> <tag> [env...]
CFG=${1:-cfg4}; TAG=${2:-base}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config $CFG --steps 60 --warmup 20 --no-cpu-baseline --no-micro --profile-steps 0 --no-graph"
rm -rf /tmp/q1 /tmp/q2 /tmp/q3
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/q1 -- $B > $OUT/q1.log 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d /tmp/q2 -- $B > $OUT/q2.log 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d /tmp/q3 -- $B > $OUT/q3.log 2>&1
python $ROOT/tools/pmc_counters.py $OUT/stall_${CFG}.json /tmp/q1/*/*counter_collection.csv /tmp/q2/*/*counter_collection.csv /tmp/q3/*/*counter_collection.csv
