#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests -q -m gpu --timeout 200 -p no:cacheprovider > gpurun_out/r3_full2.log 2>&1; tail -4 gpurun_out/r3_full2.log
for k in 12 20 30; do echo "== cfg4 KSLABS=$k"; G4R_KSLABS=$k timeout 120 python bench.py --config cfg4 --steps 1000 --warmup 200 --no-cpu-baseline --no-micro --long-steps 0 > gpurun_out/r3_ab.json 2> gpurun_out/r3_ab.err; python tools/benchsum.py gpurun_out/r3_ab.json; done
timeout 500 bash tools/final_profile.sh r03 "cfg3 cfg4" > gpurun_out/final_profile_r03b.log 2>&1; grep -A4 "^== cfg[34]$" gpurun_out/final_profile_r03b.log | cut -c1-200 | head -40
