#!/bin/bash
# A/B of gemm_tile2k pipelines: committed ("old"), two-ahead at 4 waves/SIMD (product build), two-ahead at 5 waves/SIMD
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_dma_tiles.py tests/test_gpu_baseline_configs.py tests/test_gpu_catalogue_scale.py -q -m gpu -x --timeout 120 -p no:cacheprovider > gpurun_out/r3_ab2_tests.log 2>&1; tail -3 gpurun_out/r3_ab2_tests.log
for v in old main w5 old main w5; do
  for c in cfg4 cfg3; do
    lib=""; [ $v != main ] && lib="gru4rec_amd/_variants/libgru4rec_hip_$v.so"
    echo "== $v $c"
    G4R_LIB=$lib timeout 100 python bench.py --config $c --steps 1000 --warmup 200 --no-cpu-baseline --no-micro --long-steps 0 > gpurun_out/r3_ab2_${v}_$c.json 2> gpurun_out/r3_ab2.err
    python tools/benchsum.py gpurun_out/r3_ab2_${v}_$c.json
  done
done
