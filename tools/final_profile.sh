#!/bin/bash
# Everything the round's profile files come from, in one GPU call:  bash tools/final_profile.sh <round tag, e.g. r04> ["cfg2 cfg3 cfg4"]
# Every command runs under its own timeout.  Order: the bench lines first (default line with the CPU baseline; the other configs;
# the N > 1 data path with a one-rank communicator in both item-row modes), then per config one rocprofv3 --kernel-trace --stats
# pass of the graph-replay run (eager if the profiler cannot follow the graph), then two PMC passes (FETCH_SIZE, WRITE_SIZE: they
# cannot share a pass on gfx950) -> per-kernel HBM-side traffic (tools/pmc_summary.py) -- the second pair with G4R_NO_MERGE=1, so
# that the sparse row update (k_sparse_update) has counters of its own --, then the MFMA-busy counters of the DEFAULT kernels.
# rocprofv3 runs from /tmp as the guide prescribes.  Outputs: gpurun_out/final/ (copy the ones to keep into profiles/).
TAG=${1:-r05}
CFGS=${2:-"cfg2 cfg3 cfg4"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
B="python $ROOT/bench.py"
cd $ROOT
python -c "from gru4rec_amd import _native; print(_native.lib().g4r_version().decode())" > $OUT/${TAG}_library_version.txt
if [ -z "$PMC_ONLY" ]; then      # PMC_ONLY=1: only the counter passes at the end of this script
timeout 300 $B > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err; python tools/benchsum.py $OUT/${TAG}_bench_default.json
timeout 120 $B --steps 20 --warmup 5 --no-micro > $OUT/${TAG}_bench_driver_shape.json 2> $OUT/bench_driver_shape.err; python tools/benchsum.py $OUT/${TAG}_bench_driver_shape.json
for c in cfg3 cfg4 cfg1 cfg5; do timeout 240 $B --config $c --steps 1500 --warmup 200 --no-cpu-baseline > $OUT/${TAG}_bench_$c.json 2> $OUT/bench_$c.err; echo "== $c"; python tools/benchsum.py $OUT/${TAG}_bench_$c.json; done
# deferred row updates (opt-in): the flush launch of every window priced in roofline_gather_scatter.deferred_flush
for c in cfg2 cfg3 cfg4; do timeout 240 $B --config $c --steps 1500 --warmup 200 --no-cpu-baseline --no-micro --defer > $OUT/${TAG}_bench_${c}_defer.json 2> $OUT/bench_${c}_defer.err; echo "== $c, deferred row updates"; python tools/benchsum.py $OUT/${TAG}_bench_${c}_defer.json; done
G4R_FORCE_STAGED=1 timeout 200 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > $OUT/${TAG}_bench_staged_1rank.json 2> $OUT/bench_staged.err; echo "== staged, gpu-local rows"; python tools/benchsum.py $OUT/${TAG}_bench_staged_1rank.json
G4R_FORCE_STAGED=1 timeout 200 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro --sparse-exact > $OUT/${TAG}_bench_staged_1rank_exact.json 2> $OUT/bench_staged_exact.err; echo "== staged, exact replicas"; python tools/benchsum.py $OUT/${TAG}_bench_staged_1rank_exact.json
G4R_FORCE_STAGED=1 timeout 240 $B --config cfg4 --steps 600 --warmup 100 --no-cpu-baseline --no-micro > $OUT/${TAG}_bench_staged_1rank_cfg4.json 2> $OUT/bench_staged_cfg4.err; echo "== staged cfg4"; python tools/benchsum.py $OUT/${TAG}_bench_staged_1rank_cfg4.json
G4R_FORCE_STAGED=1 timeout 240 $B --config cfg4 --steps 600 --warmup 100 --no-cpu-baseline --no-micro --sparse-exact > $OUT/${TAG}_bench_staged_1rank_cfg4_exact.json 2> $OUT/bench_staged_cfg4_exact.err; echo "== staged cfg4, exact replicas"; python tools/benchsum.py $OUT/${TAG}_bench_staged_1rank_cfg4_exact.json
fi
[ -n "$BENCH_ONLY" ] && exit 0      # BENCH_ONLY=1: the bench lines alone (they read the committed profiles/ traffic files)
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0"
[ -n "$PMC_ONLY" ] && SKIP_STATS=1
for c in $CFGS; do
  [ -n "$SKIP_STATS" ] && break
  rm -rf /tmp/out_s
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --config $c --steps 600 --warmup 100 $COMMON > $OUT/stats_$c.log 2>&1
  if ! ls /tmp/out_s/*/*kernel_stats.csv > /dev/null 2>&1; then
    echo "($c: the profiler did not survive the graph replay; eager launches)"; rm -rf /tmp/out_s
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --config $c --steps 600 --warmup 100 --no-graph $COMMON > $OUT/stats_$c.log 2>&1
  fi
  cp /tmp/out_s/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_rocprofv3_$c.csv 2>/dev/null
  echo "== $c"; head -12 $OUT/${TAG}_kernel_stats_rocprofv3_$c.csv
done
# the same with deferred row updates at the two big-catalogue configs: k_sparse_flush on the profiler's clock and under the traffic counters
for c in cfg3 cfg4; do
  [ -n "$SKIP_STATS" ] && break
  rm -rf /tmp/out_s
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --config $c --steps 600 --warmup 100 --defer $COMMON > $OUT/stats_${c}_defer.log 2>&1
  cp /tmp/out_s/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_rocprofv3_${c}_defer.csv 2>/dev/null
  echo "== $c (deferred row updates)"; grep -i "flush\|defer_scan\|k_update\|k_sparse_update" $OUT/${TAG}_kernel_stats_rocprofv3_${c}_defer.csv | head -6
done
[ -n "$STATS_ONLY" ] && exit 0      # STATS_ONLY=1: bench lines + rocprofv3 kernel statistics, no counter passes
for c in cfg3 cfg4; do
  rm -rf /tmp/out_f /tmp/out_w
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- $B --config $c --steps 128 --warmup 32 --defer --no-graph $COMMON > $OUT/pmc_f_${c}_defer.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- $B --config $c --steps 128 --warmup 32 --defer --no-graph $COMMON > $OUT/pmc_w_${c}_defer.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $OUT/${TAG}_pmc_traffic_${c}_defer.json > $OUT/pmc_summary_${c}_defer.txt 2>&1
  echo "== $c traffic (deferred row updates)"; grep -i "flush\|k_update\|k_sparse_update\|defer" $OUT/pmc_summary_${c}_defer.txt
done
for c in $CFGS; do
  for nm in merged split; do
    [ $nm = split ] && [ $c = cfg3 ] && continue      # (configs[2]: the update already runs as two launches)
    rm -rf /tmp/out_f /tmp/out_w
    if [ $nm = split ]; then export G4R_NO_MERGE=1; else unset G4R_NO_MERGE; fi
    timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- $B --config $c --steps 100 --warmup 20 --no-graph $COMMON > $OUT/pmc_f_$c.log 2>&1
    timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- $B --config $c --steps 100 --warmup 20 --no-graph $COMMON > $OUT/pmc_w_$c.log 2>&1
    sfx=""; [ $nm = split ] && sfx="_no_merge"
    python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $OUT/${TAG}_pmc_traffic_$c$sfx.json > $OUT/pmc_summary_$c$sfx.txt 2>&1
    echo "== $c traffic ($nm)"; cat $OUT/pmc_summary_$c$sfx.txt
  done
  unset G4R_NO_MERGE
done
for c in cfg3 cfg4; do
  rm -rf /tmp/p1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/p1 -- $B --config $c --steps 60 --warmup 20 --no-graph $COMMON > $OUT/pmc_mfma_$c.log 2>&1
  python $ROOT/tools/pmc_mfma_summary.py $OUT/${TAG}_pmc_mfma_$c.json /tmp/p1 $c 2>&1 | head -14
done
