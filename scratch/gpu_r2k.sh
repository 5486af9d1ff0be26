cd $GRAFT_REPO_ROOT
for l in w2 w3; do TINSEL_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libtinsel_hip_$l.so python scratch/fast_cmp.py 2>&1 | grep -v amdgpu; done
