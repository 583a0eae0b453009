#!/bin/bash
# usage: tools/bench_variants.sh <tag> "<cfgs>" "<env variants, ';'-separated, '-' = none>"
TAG=$1; CFGS=${2:-"cfg3 cfg4"}; VARS=${3:-"-"}
mkdir -p gpurun_out/$TAG
IFS=';' read -ra VV <<< "$VARS"
for c in $CFGS; do
  for v in "${VV[@]}"; do
    name=$(echo "$v" | tr ' =' '__'); [ "$v" = "-" ] && name=base
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    env $envs timeout 300 python bench.py --config $c --steps 1500 --warmup 200 --no-cpu-baseline > gpurun_out/$TAG/${c}_$name.json 2> gpurun_out/$TAG/${c}_$name.err
    echo "== $c $v"; python tools/benchsum.py gpurun_out/$TAG/${c}_$name.json
  done
done
