#!/bin/bash
# Counter passes of the deferred-update run alone (k_sparse_flush under FETCH_SIZE / WRITE_SIZE):  bash tools/pmc_defer.sh r05 "cfg3 cfg4"
TAG=${1:-r05}; CFGS=${2:-"cfg3 cfg4"}
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT; B="python $ROOT/bench.py"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0"
for c in $CFGS; do
  rm -rf /tmp/out_f /tmp/out_w
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- $B --config $c --steps 256 --warmup 64 --defer --no-graph $COMMON > $OUT/pmc_f_${c}_defer.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- $B --config $c --steps 256 --warmup 64 --defer --no-graph $COMMON > $OUT/pmc_w_${c}_defer.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $OUT/${TAG}_pmc_traffic_${c}_defer.json > $OUT/pmc_summary_${c}_defer.txt 2>&1
  echo "== $c traffic (deferred row updates)"; grep -i "flush\|k_update\|k_sparse_update\|defer" $OUT/pmc_summary_${c}_defer.txt
done
