#!/bin/bash
# A/B of experiment builds (build/exp/libphaze_<name>.so) against the product on ONE box, product first and last:
#   tools/ab.sh <outdir> <shapes> [steps] -- name ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/$1; SHAPES=$2; shift 2; STEPS=10
if [ "$1" != "--" ]; then STEPS=$1; shift; fi; shift
mkdir -p $OUT
{
python tools/ab_shapes.py product $SHAPES $STEPS
for n in "$@"; do PHAZE_LIB=$ROOT/build/exp/libphaze_$n.so python tools/ab_shapes.py $n $SHAPES $STEPS; done
python tools/ab_shapes.py product2 $SHAPES $STEPS
} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
