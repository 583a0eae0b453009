#!/bin/bash
# A/B of library variants on ONE box: [CFG=cfg4] [REPS=2] [STEPS=3000] tools/ab.sh lib_base.so lib_x.so ...   (files under tmp_var/)
cd "$(dirname "$0")/.."
for rep in $(seq ${REPS:-2}); do
for v in "$@"; do
  G4R_LIB=$PWD/tmp_var/$v timeout 200 python bench.py --config ${CFG:-cfg2} --steps ${STEPS:-3000} --warmup 300 --no-cpu-baseline > /tmp/ab_$v.log 2>&1
  echo "== $v"; python tools/benchsum.py /tmp/ab_$v.log
done
done
