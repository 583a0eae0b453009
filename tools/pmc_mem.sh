#!/bin/bash
# memory-path counters of the step kernels (L1 -> L2 requests and their latency, L2 hit rate, translation misses):
#   bash tools/pmc_mem.sh <cfg> <tag> [env...]
CFG=${1:-cfg4}; TAG=${2:-base}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config $CFG --steps 60 --warmup 20 --no-cpu-baseline --no-micro --profile-steps 0 --no-graph"
rm -rf /tmp/m1 /tmp/m2 /tmp/m3
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/m1 -- $B > $OUT/m1.log 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum --output-format csv -d /tmp/m2 -- $B > $OUT/m2.log 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/m3 -- $B > $OUT/m3.log 2>&1
python $ROOT/tools/pmc_counters.py $OUT/mem_${CFG}.json /tmp/m1/*/*counter_collection.csv /tmp/m2/*/*counter_collection.csv /tmp/m3/*/*counter_collection.csv
