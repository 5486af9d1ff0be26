#!/bin/bash
# prints per-kernel VGPR/SGPR/scratch/occupancy of the HIP library
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math "$@" -Rpass-analysis=kernel-resource-usage -o /tmp/x.so /root/repo/tinsel_amd/csrc/tinsel_hip.hip 2>&1 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | python3 -c "
import sys,re
name=None; d={}
for l in sys.stdin:
    l=l.strip()
    if l.startswith('Function Name:'):
        name=l.split(':',1)[1].strip(); d={}
    elif ':' in l:
        k,v=l.split(':',1); d[k.strip()]=v.strip()
        if k.strip().startswith('LDS Size'):
            n=re.sub(r'^_ZN2tn\d+','',name); n=re.sub(r'(ILb[01]E(Lb[01]E)?)?E?vNS_.*|ENS_.*','',n) + (' '+''.join(re.findall(r'Lb([01])',name)) if 'ILb' in name else '')
            print('%-22s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %s' % (n, d.get('VGPRs'), d.get('AGPRs'), d.get('SGPRs'), d.get('ScratchSize [bytes/lane]'), d.get('Occupancy [waves/SIMD]')))
"
