#!/bin/bash
# registers, scratch, LDS and occupancy of every step kernel of the module (hipcc's own remarks), one line per instantiation
# usage: tools/resource_usage.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null datum_amd/csrc/ocean_capi.hip 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r' SGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=int(m.group(1))
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=n.replace('ocean::','').replace('(StepArgs)','')
    if 'pass' in n or 'gen' in n:
        print(f\"{n:<55} vgpr {r.get('vgpr',0):4d} sgpr {r.get('sgpr',0):4d} scratch {r.get('scratch',0):4d} occ {r.get('occ',0)} lds(static) {r.get('lds',0)}\")
"
