#!/bin/bash
# kernel resource usage of one csrc file: tools/kres.sh radix_sort.hip [extra flags]  -> name vgprs agprs sgprs lds scratch
f=$1; shift
cd /root/repo/a-recsys_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I. "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass',l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'): cur={'name':t.split(': ',1)[1]}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=n.replace('arx::(anonymous namespace)::','').replace('arx::','').replace('void ','');n=re.sub(r'\(.*','',n)
    print('%-48s vgpr %4s agpr %4s sgpr %4s lds %6s scratch %s occ %s'%(n[:48],r.get('VGPRs'),r.get('AGPRs'),r.get('TotalSGPRs'),r.get('LDS Size [bytes/block]'),r.get('ScratchSize [bytes/lane]'),r.get('Occupancy [waves/SIMD]')))
"
