#!/bin/bash
# per-kernel register / LDS / spill figures (compiler remarks of a device-only compile of the library source)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c gru4rec_amd/csrc/g4r_api.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z][^:]*?): (\S+)',line)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0].replace('void ','')
    print('%-44s vgpr %4s agpr %3s sgpr %4s spillV %3s spillS %3s scratch %5s occ %2s lds %6s' % (name[:44], v.get('VGPRs'), v.get('AGPRs'), v.get('TotalSGPRs'), v.get('VGPRs Spill'), v.get('SGPRs Spill'), v.get('ScratchSize [bytes/lane]'), v.get('Occupancy [waves/SIMD]'), v.get('LDS Size [bytes/block]')))
"
