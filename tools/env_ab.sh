# one-off A/B of an environment switch on the bench lines:  bash tools/sf_ab.sh VAR "v1 v2 ..." "cfg3 cfg4" [pytest args]
set -x
VAR=$1; VALS=$2; CFGS=${3:-cfg3}; OUT=gpurun_out/envab; mkdir -p $OUT
if [ -n "$4" ]; then timeout 900 python -m pytest $4 -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log; fi
for rep in 1 2; do
for c in $CFGS; do
for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --config $c --steps 1000 --warmup 300 --no-cpu-baseline --no-micro > $OUT/${c}_${VAR}${v}_$rep.json 2> $OUT/${c}_${VAR}${v}_$rep.err
  echo "== $c $VAR=$v rep $rep"; python tools/benchsum.py $OUT/${c}_${VAR}${v}_$rep.json
done
done
done
