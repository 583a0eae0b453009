#!/bin/bash
# Round 3, last GPU call: the whole GPU suite (4 workers, one file per worker at a time), smoke, the default bench line, the cfg #3 / #4
# bench lines with and without the compact copy of the score rows (G4R_SYC), a rocprofv3 kernel-stats pass at cfg #4, cfg4s.
cd "$(dirname "$0")/.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final2; mkdir -p $OUT
T0=$(date +%s)
timeout 330 python -m pytest tests -q -m gpu -n 4 --dist loadfile --timeout 250 -p no:cacheprovider > $OUT/tests.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -6 $OUT/tests.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 150 python bench.py > $OUT/r03_bench_default.json 2> $OUT/bench_default.err; python tools/benchsum.py $OUT/r03_bench_default.json
for c in cfg4 cfg3; do for v in 1 0; do
  echo "== $c G4R_SYC=$v"
  G4R_SYC=$v timeout 100 python bench.py --config $c --steps 1500 --warmup 200 --no-cpu-baseline --long-steps 0 > $OUT/r03_bench_${c}_syc$v.json 2> $OUT/bench_$c.err
  python tools/benchsum.py $OUT/r03_bench_${c}_syc$v.json
done; done
echo "== cfg4 G4R_SYC=2 (forked copy)"; G4R_SYC=2 timeout 100 python bench.py --config cfg4 --steps 1500 --warmup 200 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/r03_bench_cfg4_syc2.json 2>> $OUT/bench_cfg4.err; python tools/benchsum.py $OUT/r03_bench_cfg4_syc2.json
echo "== cfg4s"; timeout 100 python bench.py --config cfg4s --steps 1500 --warmup 200 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/r03_bench_cfg4s.json 2>> $OUT/bench_cfg4.err; python tools/benchsum.py $OUT/r03_bench_cfg4s.json
echo "elapsed $(( $(date +%s) - T0 )) s"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/out_s
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- python $ROOT/bench.py --config cfg4 --steps 600 --warmup 100 --no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0 > $OUT/stats_cfg4.log 2>&1
cp /tmp/out_s/*/*kernel_stats.csv $OUT/r03_kernel_stats_rocprofv3_cfg4.csv 2>/dev/null; head -12 $OUT/r03_kernel_stats_rocprofv3_cfg4.csv | cut -c1-160
echo "elapsed $(( $(date +%s) - T0 )) s"
if [ $(( $(date +%s) - T0 )) -lt 360 ]; then
  rm -rf /tmp/out_s
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- python $ROOT/bench.py --config cfg3 --steps 600 --warmup 100 --no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0 > $OUT/stats_cfg3.log 2>&1
  cp /tmp/out_s/*/*kernel_stats.csv $OUT/r03_kernel_stats_rocprofv3_cfg3.csv 2>/dev/null; head -12 $OUT/r03_kernel_stats_rocprofv3_cfg3.csv | cut -c1-160
  echo "elapsed $(( $(date +%s) - T0 )) s"
fi
if [ $(( $(date +%s) - T0 )) -lt 380 ]; then
  rm -rf /tmp/out_f /tmp/out_w
  COMMON="--no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0"
  timeout 45 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- python $ROOT/bench.py --config cfg4 --steps 100 --warmup 20 --no-graph $COMMON > $OUT/pmc_f_cfg4.log 2>&1
  timeout 45 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- python $ROOT/bench.py --config cfg4 --steps 100 --warmup 20 --no-graph $COMMON > $OUT/pmc_w_cfg4.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $OUT/r03_pmc_traffic_cfg4.json > $OUT/pmc_summary_cfg4.txt 2>&1
  echo "== cfg4 traffic"; cat $OUT/pmc_summary_cfg4.txt | head -20
  echo "elapsed $(( $(date +%s) - T0 )) s"
fi
