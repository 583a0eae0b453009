#!/bin/bash
# Round 5 probe: do the wide-layer tiles get faster with more operand bytes in flight?  k_gru_bwd_aw / k_gru_bwd_bw (gemm_tile3: LDS-DMA
# ring) at ring depths 3 / 6 / 9 (variant libraries tmp_var/lib_nst{6,9}.so), with and without K slices.
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
run() {  # name cfg steps env...
  name=$1; c=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --config $c --steps $steps --warmup 100 --no-cpu-baseline --no-micro --long-steps 0 > $OUT/pr_${name}_${c}.json 2> $OUT/pr_${name}_${c}.err
  echo "== $name $c"; python tools/benchsum.py $OUT/pr_${name}_${c}.json
}
for lib in base nst6 nst9; do
  L=""; [ $lib != base ] && L="G4R_LIB=$ROOT/tmp_var/lib_$lib.so"
  for ks in 192 384 1536; do
    run ${lib}_ks$ks cfg3 300 G4R_WIDE2=12 G4R_BB_KS=$ks G4R_BA_KS=$([ $ks = 1536 ] && echo 512 || echo 128) $L
  done
done
