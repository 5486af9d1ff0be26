#!/bin/bash
# round 5, call c: k_walk with two tree levels per cache line (Fat128 / kWalkFat) under the whole suite and against the plain walk; the split
# pipeline's kernels of this tree against the library of call a (are k_shade / k_extend slower, or was it the box?); glass per launch; counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1; grep -a "passed\|failed\|rror" $O/pytest_gpu.log | tail -8
{
echo "| environment | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_FAT=0" "-" "TINSEL_HIP_WALK_FAT=0" -- --scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_FAT=0" -- --scene motionblur --width 1024 --height 1024 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_FAT=0" -- --scene large/table --width 1024 --height 1024 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_FAT=0" -- --scene large/transmission --width 1024 --height 1024 --steps 20 --warmup 5
bash scratch/gpu_envs.sh "-" "TINSEL_HIP_WALK_FAT=0" -- --scene ajax_standin_96 --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5
} > $O/ab_walk_fat.md 2>&1; cat $O/ab_walk_fat.md
{
echo "| library | workload | Msamples/s | kernel ms |"; echo "|---|---|---|---|"
for W in "--scene glass --width 1920 --height 1080 --maxdepth 12 --steps 20 --warmup 5" "--scene many_spheres --width 1024 --height 768 --steps 64 --warmup 8" "--scene large/ajax_standin --width 1920 --height 1080 --maxdepth 4 --steps 20 --warmup 5"; do
  bash scratch/gpu_envs.sh "TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_r5a.so" "TINSEL_HIP_WALK_FAT=0" "TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/libtinsel_hip_r5a.so" "TINSEL_HIP_WALK_FAT=0" -- $W
done
} > $O/ab_split_vs_call_a.md 2>&1; cat $O/ab_split_vs_call_a.md
( export TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_walkprof.so
  python scratch/walk_prof.py large/ajax_standin 1920 1080 4 20; TINSEL_HIP_WALK_FAT=0 python scratch/walk_prof.py large/ajax_standin 1920 1080 4 20 ) > $O/walk_profile_cfg3.txt 2>&1; cat $O/walk_profile_cfg3.txt
bash scratch/gpu_pmc_kernels.sh $O fat "large/ajax_standin 1920 1080 4 20" "glass 1920 1080 12 20" > /dev/null 2>&1; cat $O/pmc_fat.md
# glass, one batch, launch by launch
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o tl_glass --output-format csv -- python $GRAFT_REPO_ROOT/scratch/launch_timeline.py glass 1920 1080 12 20 2>/dev/null | grep "K =" > $O/timeline_glass.txt
python $GRAFT_REPO_ROOT/scratch/launch_timeline.py --parse $(find $O -name "tl_glass*kernel_trace.csv" | head -1) >> $O/timeline_glass.txt; cat $O/timeline_glass.txt
find $O -name "*.csv" -delete
