#!/bin/bash
# call 2d: fewer dependent fetches in the flat scan, continued: the plane table's tail requested with its first block; a quad's node / triangles /
# normals found through offsets in its primitive record (no mesh-table record in between); both of a quad's triangles requested before either test
O=gpurun_out/r6_2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_refit.py tests/test_gpu_leaf.py tests/test_gpu_configs.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 2400 python scratch/ab_rates.py --repeat 3 --lib base=scratch/ab/libtinsel_hip_base.so --lib ahead=scratch/ab/libtinsel_hip_ahead.so --lib tail=scratch/ab/libtinsel_hip_tail.so --lib quadrec=tinsel_amd/libtinsel_hip.so --lib tris=scratch/ab/libtinsel_hip_tris.so \
    cornell veach4k cfg1 > $O/ab_scan2.md 2> $O/ab.err; cat $O/ab_scan2.md
timeout 1200 python scratch/ab_rates.py --repeat 3 --lib 'paired=args:--pipeline paired' --lib 'split=args:--pipeline split' --lib 'split_tris=scratch/ab/libtinsel_hip_tris.so;args:--pipeline split' glass > $O/ab_glass.md 2>> $O/ab.err; cat $O/ab_glass.md
