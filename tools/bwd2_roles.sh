#!/bin/bash
# k_score_bwd2 one role at a time (G4R_DEBUG_BWD2_ROLE: the training results are wrong, only the launch is of interest): duration from
# bench.py's per-kernel HIP events, then two PMC passes (MFMA / LDS) per role.   bash tools/bwd2_roles.sh [cfg4]
CFG=${1:-cfg4}
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/bwd2_roles; mkdir -p $OUT
for r in 0 1 2; do
  echo "== role $r (0 = the whole launch, 1 = dS tiles, 2 = dh slabs)"
  G4R_DEBUG_BWD2_ROLE=$r timeout 60 python $ROOT/bench.py --config $CFG --steps 600 --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/bench_role$r.json 2> $OUT/bench_role$r.err
  python $ROOT/tools/benchsum.py $OUT/bench_role$r.json | tail -1
done
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config $CFG --steps 60 --warmup 20 --no-cpu-baseline --no-micro --profile-steps 0 --long-steps 0 --no-graph"
for r in 1 2; do
  rm -rf /tmp/p1 /tmp/p2
  G4R_DEBUG_BWD2_ROLE=$r timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/p1 -- $B > $OUT/p1_role$r.log 2>&1
  G4R_DEBUG_BWD2_ROLE=$r timeout 90 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/p2 -- $B > $OUT/p2_role$r.log 2>&1
  python $ROOT/tools/pmc_counters.py $OUT/counters_role$r.json /tmp/p1/*/*counter_collection.csv /tmp/p2/*/*counter_collection.csv 2>&1 | grep -i "bwd2" | head -4
done
