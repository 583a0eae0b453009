#!/bin/bash
# Everything the round's profile files come from, in one GPU call:  bash tools/final_profile.sh <round tag, e.g. r02> ["cfg2 cfg3 cfg4"]
# Per config: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they cannot share a pass on gfx950) -> per-kernel HBM-side traffic
# (tools/pmc_summary.py), one rocprofv3 --kernel-trace --stats pass of the graph-replay run -> kernel_stats CSV, then the bench line
# itself (which reads the traffic file).  rocprofv3 runs from /tmp as the guide prescribes.  Outputs: gpurun_out/final/ (copy the
# ones to keep into profiles/).
TAG=${1:-r02}
CFGS=${2:-"cfg2 cfg3 cfg4"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py"
for c in $CFGS; do
  ST=300; [ "$c" = "cfg2" ] || ST=150
  rm -rf /tmp/out_f /tmp/out_w /tmp/out_s
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- $B --config $c --steps $ST --warmup 40 --no-cpu-baseline --no-micro --profile-steps 0 --no-graph > $OUT/pmc_f_$c.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- $B --config $c --steps $ST --warmup 40 --no-cpu-baseline --no-micro --profile-steps 0 --no-graph > $OUT/pmc_w_$c.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $ROOT/profiles/${TAG}_pmc_traffic_$c.json > $OUT/pmc_summary_$c.txt 2>&1
  cp $ROOT/profiles/${TAG}_pmc_traffic_$c.json $OUT/ 2>/dev/null
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --config $c --steps 1000 --warmup 150 --no-cpu-baseline --no-micro > $OUT/stats_$c.log 2>&1
  cp /tmp/out_s/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_rocprofv3_$c.csv 2>/dev/null
  echo "== $c"; cat $OUT/pmc_summary_$c.txt; head -14 $OUT/${TAG}_kernel_stats_rocprofv3_$c.csv
done
cd $ROOT
timeout 500 $B > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
for c in cfg1 cfg3 cfg4 cfg5; do timeout 400 $B --config $c --steps 1500 --warmup 200 --no-cpu-baseline > $OUT/${TAG}_bench_$c.json 2> $OUT/bench_$c.err; done
G4R_FORCE_STAGED=1 timeout 300 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > $OUT/${TAG}_bench_staged_1rank.json 2> $OUT/bench_staged.err
for f in $OUT/${TAG}_bench_*.json; do echo "== $f"; python tools/benchsum.py $f; done
