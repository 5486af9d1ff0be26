#!/bin/bash
# prints per-kernel VGPR/SGPR/scratch/occupancy of the HIP library: `resusage.sh` the parity build, `resusage.sh fast` the tolerance arm
if [ "$1" = "fast" ]; then shift; SRC=/root/repo/tinsel_amd/csrc/tinsel_fast.hip; FLAGS="-DTN_FAST=1 -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fgpu-flush-denormals-to-zero"
else SRC=/root/repo/tinsel_amd/csrc/tinsel_hip.hip; FLAGS="-ffp-contract=off -fno-fast-math"; fi
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $FLAGS "$@" --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o /tmp/x.o $SRC 2>&1 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | python3 -c "
import sys,re
name=None; d={}
for l in sys.stdin:
    l=l.strip()
    if l.startswith('Function Name:'):
        name=l.split(':',1)[1].strip(); d={}
    elif ':' in l:
        k,v=l.split(':',1); d[k.strip()]=v.strip()
        if k.strip().startswith('LDS Size'):
            n=re.sub(r'^_ZN[0-9a-z_]*?(\d+)(k_)', r'\2', name); n=re.sub(r'(ILb[01]E(Lb[01]E)*)?E?vNS_.*|ENS_.*','',n) + (' '+''.join(re.findall(r'Lb([01])',name)) if 'ILb' in name else '')
            print('%-26s VGPR %4s AGPR %3s scratch %5s occ %s' % (n, d.get('VGPRs'), d.get('AGPRs'), d.get('ScratchSize [bytes/lane]'), d.get('Occupancy [waves/SIMD]')))
"
