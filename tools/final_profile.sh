#!/bin/bash
# Everything the round's profile files come from, in one GPU call:  bash tools/final_profile.sh <tag>
# (PMC passes first: bench.py reads the traffic file they produce; rocprofv3 runs from /tmp as the guide prescribes)
TAG=${1:-v7}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/out_f -- $B --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-graph > $OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/out_w -- $B --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-graph > $OUT/pmc_w.log 2>&1
python $ROOT/tools/pmc_summary.py /tmp/out_f/*/*counter_collection.csv /tmp/out_w/*/*counter_collection.csv $ROOT/profiles/r01_pmc_traffic_cfg2.json > $OUT/pmc_summary.txt 2>&1
cp $ROOT/profiles/r01_pmc_traffic_cfg2.json $OUT/
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/out_s -- $B --steps 1200 --warmup 200 --no-cpu-baseline > $OUT/stats.log 2>&1
cp /tmp/out_s/*/*kernel_stats.csv $OUT/r01_${TAG}_kernel_stats_rocprofv3_cfg2.csv 2>/dev/null
cd $ROOT
timeout 400 $B > $OUT/r01_bench_${TAG}_default.json 2> $OUT/bench_default.err
for c in cfg1 cfg3 cfg4 cfg5; do timeout 300 $B --config $c --steps 1500 --warmup 200 --no-cpu-baseline > $OUT/r01_bench_${TAG}_$c.json 2> $OUT/bench_$c.err; done
G4R_FORCE_STAGED=1 timeout 300 $B --steps 3000 --warmup 300 --no-cpu-baseline > $OUT/r01_bench_${TAG}_staged_1rank.json 2> $OUT/bench_staged.err
for f in $OUT/r01_bench_${TAG}_*.json; do echo "== $f"; python tools/benchsum.py $f; done
cat $OUT/pmc_summary.txt
head -12 $OUT/r01_${TAG}_kernel_stats_rocprofv3_cfg2.csv
