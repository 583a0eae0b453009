#!/bin/bash
# Everything the round's profile files come from, in one GPU call:  bash tools/final_profile.sh <round tag, e.g. r03> ["cfg2 cfg3 cfg4"]
# Every command runs under its own timeout.  Order: the bench lines first, then per config one rocprofv3 --kernel-trace --stats pass
# of the graph-replay run (eager if the profiler cannot follow the graph), then two PMC passes (FETCH_SIZE, WRITE_SIZE: they cannot
# share a pass on gfx950) -> per-kernel HBM-side traffic (tools/pmc_summary.py).  rocprofv3 runs from /tmp as the guide prescribes.
# Outputs: gpurun_out/final/ (copy the ones to keep into profiles/).
TAG=${1:-r03}
CFGS=${2:-"cfg2 cfg3 cfg4"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
B="python $ROOT/bench.py"
cd $ROOT
timeout 240 $B > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err; python tools/benchsum.py $OUT/${TAG}_bench_default.json
for c in cfg3 cfg4 cfg1 cfg5; do timeout 150 $B --config $c --steps 1500 --warmup 200 --no-cpu-baseline > $OUT/${TAG}_bench_$c.json 2> $OUT/bench_$c.err; echo "== $c"; python tools/benchsum.py $OUT/${TAG}_bench_$c.json; done
G4R_FORCE_STAGED=1 timeout 150 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > $OUT/${TAG}_bench_staged_1rank.json 2> $OUT/bench_staged.err; python tools/benchsum.py $OUT/${TAG}_bench_staged_1rank.json
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0"
for c in $CFGS; do
  rm -rf /tmp/out_s
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --config $c --steps 600 --warmup 100 $COMMON > $OUT/stats_$c.log 2>&1
  if ! ls /tmp/out_s/*/*kernel_stats.csv > /dev/null 2>&1; then
    echo "($c: the profiler did not survive the graph replay; eager launches)"; rm -rf /tmp/out_s
    timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --config $c --steps 600 --warmup 100 --no-graph $COMMON > $OUT/stats_$c.log 2>&1
  fi
  cp /tmp/out_s/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_rocprofv3_$c.csv 2>/dev/null
  echo "== $c"; head -12 $OUT/${TAG}_kernel_stats_rocprofv3_$c.csv
done
for c in $CFGS; do
  rm -rf /tmp/out_f /tmp/out_w
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- $B --config $c --steps 100 --warmup 20 --no-graph $COMMON > $OUT/pmc_f_$c.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- $B --config $c --steps 100 --warmup 20 --no-graph $COMMON > $OUT/pmc_w_$c.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $OUT/${TAG}_pmc_traffic_$c.json > $OUT/pmc_summary_$c.txt 2>&1
  echo "== $c traffic"; cat $OUT/pmc_summary_$c.txt
done
