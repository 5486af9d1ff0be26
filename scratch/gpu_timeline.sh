#!/bin/bash
# per-launch timeline of one batch for a few configs; usage: gpu_timeline.sh [lib ...] (default: the in-tree build)
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tl; mkdir -p $O
for L in ${@:-tinsel_amd/libtinsel_hip.so}; do
export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/$L
T=$(basename $L .so)
for cfg in "glass 1920 1080 12 32" "veach 3840 2160 4 8" "many_spheres 1024 768 4 64" "large/ajax_standin 1920 1080 4 32"; do
set -- $cfg
N=${T}_$(basename $1)
echo "== $L $cfg"
timeout 300 rocprofv3 --kernel-trace -d $O -o tl_$N --output-format csv -- python $GRAFT_REPO_ROOT/scratch/launch_timeline.py $cfg 2>/dev/null | grep "K =" | tee $O/counts_$N.txt
python $GRAFT_REPO_ROOT/scratch/launch_timeline.py --parse $O/tl_${N}_kernel_trace.csv | tee $O/tl_$N.txt
done
done
find $O -name "*.csv" -delete
