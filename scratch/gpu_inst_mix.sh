#!/bin/bash
# gpu_inst_mix.sh OUTDIR "scene W H depth passes" ...   dynamic instruction mix per kernel (two --pmc passes, scratch/inst_mix.py)
OUTD=$1; shift
cd /tmp; export TMPDIR=/tmp
mkdir -p $OUTD
for sc in "$@"; do set -- $sc
B=$(basename $1)
CMD="python $GRAFT_REPO_ROOT/bench.py --inner-pmc --no-ubench --scene $1 --width $2 --height $3 --maxdepth $4 --steps $5"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -d $OUTD/rawa_$B -o pmc --output-format csv -- $CMD > /dev/null 2> $OUTD/mix_a_$B.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUTD/rawb_$B -o pmc --output-format csv -- $CMD > /dev/null 2> $OUTD/mix_b_$B.err
A=$(find $OUTD/rawa_$B -name "*counter_collection.csv" | head -1); Bf=$(find $OUTD/rawb_$B -name "*counter_collection.csv" | head -1)
( echo "### $1 ${2}x$3 maxDepth $4, $5 passes"; python $GRAFT_REPO_ROOT/scratch/inst_mix.py $A $Bf ) | tee -a $OUTD/inst_mix.md
rm -rf $OUTD/rawa_$B $OUTD/rawb_$B
done
