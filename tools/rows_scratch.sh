#!/bin/bash
# where the rows kernel's scratch accesses sit (basic block + loop depth), one instantiation; extra flags: "$@"
cd /root/repo/fastertransformer4codefuse_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wno-unused-function -I../../include -DRW_FEW "$@" -S --cuda-device-only -o /tmp/rows.s kernels_rows.hip 2>/dev/null
python3 - <<'PY'
import re
lines=open('/tmp/rows.s').read().split('\n')
cur=None; d={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):\s*;(.*)',l)
    if m: cur=(m.group(1),m.group(2).strip(),i)
    if 'scratch_' in l and cur: d.setdefault(cur,[]).append(i)
tot=0
for k,v in d.items():
    tot+=len(v); print(k[0],k[1][:50],'at',k[2],'n=',len(v))
print('total scratch ops',tot,'| lines',len(lines))
for l in lines:
    if 'vgpr_spill_count' in l or 'scratch_en' in l or '.vgpr_count' in l: print(l.strip())
PY
