#!/bin/bash
# VGPRs / scratch / occupancy / LDS of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh locate_amd/csrc/conv.hip [name filter]
F=$1; PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$(dirname "$F")" -c "$F" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c "
import re,sys,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('spill',r'VGPRs Spill: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
names=subprocess.run(['/usr/bin/c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    n=re.sub(r'\(.*','',n)
    if re.search(r'$PAT',n): print('%-62s vgpr %3s agpr %3s scratch %4s spill %3s occ %s lds %6s'%(n[:62],r.get('vgpr'),r.get('agpr'),r.get('scratch'),r.get('spill'),r.get('occ'),r.get('lds')))
"
