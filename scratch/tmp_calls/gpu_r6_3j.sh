#!/bin/bash
O=gpurun_out/r6_3j; mkdir -p $O
timeout 2300 python scratch/big_parity.py 1 2>&1 | grep -v "amdgpu.ids\|^shutter\|^material" > $O/big_parity_other_scenes.txt; cat $O/big_parity_other_scenes.txt
