#!/bin/bash
# prints one line per kernel: name VGPRs SGPRs scratch occupancy LDS
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -Rpass-analysis=kernel-resource-usage -o /dev/null gsr_api.hip 2>&1 | python3 -c "
import sys,re
cur=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)',l)
    if m and cur: d[cur][m.group(1).strip()]=m.group(2)
import subprocess
for k,v in d.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0]
    print(f\"{name:45s} VGPR {v.get('VGPRs','?'):>4s} AGPR {v.get('AGPRs','?'):>3s} SGPR {v.get('SGPRs','?'):>4s} scratch {v.get('ScratchSize [bytes/lane]','?'):>5s} occ {v.get('Occupancy [waves/SIMD]','?'):>2s} LDS {v.get('LDS Size [bytes/block]','?'):>6s}\")
"
