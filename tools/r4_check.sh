#!/bin/bash
# Round-4 working call: GPU parity suite (4 worker processes on the one GPU) + short bench lines of cfg2 / cfg3 / cfg4.
#   bash tools/r4_check.sh [tag]     -> gpurun_out/r4/<tag>_*
TAG=${1:-a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd $ROOT
python -c "from gru4rec_amd import _native; print(_native.lib().g4r_version())"
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -n 4 > $OUT/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/${TAG}_gpu_tests.log
fi
B="python $ROOT/bench.py"
timeout 200 $B --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err; echo "== cfg2"; python tools/benchsum.py $OUT/${TAG}_bench_cfg2.json
for c in ${CFGS:-cfg3 cfg4}; do
  timeout 300 $B --config $c --steps 1000 --warmup 100 --no-cpu-baseline --no-micro ${BENCH_EXTRA} > $OUT/${TAG}_bench_$c.json 2> $OUT/${TAG}_bench_$c.err; echo "== $c"; python tools/benchsum.py $OUT/${TAG}_bench_$c.json; tail -3 $OUT/${TAG}_bench_$c.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/${TAG}_bench_*.json')):
    for line in open(f):
        if line.startswith('{'):
            d=json.loads(line); g=d.get('roofline_gather_scatter') or {}
            a=g.get('sparse_role_alone') or {}
            print(f.split('/')[-1], 'k_update %.2f us frac %.3f | sparse alone %.2f us frac %.3f dense alone %s' % (g.get('avg_us',0), g.get('frac',0), a.get('avg_us',0), a.get('frac',0), a.get('k_dense_grad_alone_us')))
PY
