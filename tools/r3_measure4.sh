#!/bin/bash
cd "$(dirname "$0")/.."
timeout 200 python -m pytest tests/test_gpu_dma_tiles.py tests/test_gpu_baseline_configs.py -q -m gpu -x --timeout 150 -p no:cacheprovider > gpurun_out/r3_t6.log 2>&1; tail -3 gpurun_out/r3_t6.log
run() { echo "== $*"; env "$@" timeout 150 python bench.py --config cfg4 --steps 1000 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json; }
run G4R_STREAMK=0
run G4R_SK_NST=3
run G4R_SK_NST=4
run G4R_SK_NST=3 G4R_LIB=$PWD/tmp_var/lib_loss512.so
run G4R_SK_NST=3 G4R_LIB=$PWD/tmp_var/lib_loss256.so
echo "== cfg3 default"; timeout 150 python bench.py --config cfg3 --steps 1000 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json
echo "== cfg2 loss256"; G4R_LIB=$PWD/tmp_var/lib_loss256.so timeout 150 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json
echo "== cfg2 loss512"; G4R_LIB=$PWD/tmp_var/lib_loss512.so timeout 150 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json
echo "== cfg2 default"; timeout 150 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-micro > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json
