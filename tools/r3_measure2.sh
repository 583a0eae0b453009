#!/bin/bash
# every command under its own timeout
cd "$(dirname "$0")/.."
timeout 150 python tools/sk_debug.py > gpurun_out/r3_sk_debug.txt 2>&1; tail -12 gpurun_out/r3_sk_debug.txt
timeout 600 python -m pytest tests/test_gpu_virtual_ranks.py tests/test_gpu_multirank.py tests/test_gpu_dma_tiles.py tests/test_gpu_baseline_configs.py tests/test_gpu_catalogue_scale.py tests/test_gpu_parity.py tests/test_gpu_shapes.py -q -m gpu -x --timeout 150 -p no:cacheprovider > gpurun_out/r3_t4.log 2>&1; tail -15 gpurun_out/r3_t4.log
for sk in 0 1; do for c in cfg4 cfg3; do G4R_STREAMK=$sk timeout 150 python bench.py --config $c --steps 1000 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_sk${sk}_$c.json 2> gpurun_out/r3_sk${sk}_$c.err; echo "== streamk=$sk $c"; python tools/benchsum.py gpurun_out/r3_sk${sk}_$c.json; done; done
CLK=$PWD/gru4rec_amd/_variants/libgru4rec_hip_clk.so
G4R_LIB=$CLK G4R_CLK=1 timeout 120 python tools/clk.py > gpurun_out/r3_clk_cfg2.txt 2>&1; tail -40 gpurun_out/r3_clk_cfg2.txt | cut -c1-330
G4R_STREAMK=0 G4R_LIB=$CLK G4R_CLK=1 CFG=cfg4 KERNEL=fwd timeout 120 python tools/clk_score.py > gpurun_out/r3_clk_cfg4_fwd.txt 2>&1; tail -8 gpurun_out/r3_clk_cfg4_fwd.txt
G4R_STREAMK=0 G4R_LIB=$CLK G4R_CLK=1 CFG=cfg4 KERNEL=bwd timeout 120 python tools/clk_score.py > gpurun_out/r3_clk_cfg4_bwd.txt 2>&1; tail -12 gpurun_out/r3_clk_cfg4_bwd.txt
G4R_FORCE_STAGED=1 timeout 150 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-micro > gpurun_out/r3_bench_staged.json 2> gpurun_out/r3_bench_staged.err; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r3_bench_staged.json') if l.startswith('{')][0])
    print('staged 1-rank:', d['value'], d.get('reconciliation_one_rank_communicator'))
except Exception as e: print('staged bench failed', e)
PY
