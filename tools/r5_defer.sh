#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_defer.py tests/test_gpu_wide_layers.py -x -q 2>&1 | tail -15
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/df_${name}_${c}.json 2> $OUT/df_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/df_${name}_${c}.json; python - <<PY
import json
d=json.loads([l for l in open('$OUT/df_${name}_${c}.json') if l.startswith('{')][0])
f=d.get('roofline_gather_scatter',{}).get('deferred_flush')
if f: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in f.items() if k!='note'})
PY
}
for c in cfg2 cfg3 cfg4; do
  st=1000; [ $c = cfg2 ] && st=3000
  run off $c $st G4R_DEFER=0
  run on $c $st G4R_DEFER=1
done
