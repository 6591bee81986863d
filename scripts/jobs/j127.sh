#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j127; mkdir -p $O
cd $R
for sh in L1_64_64; do
B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=3000 SHAPES=$sh timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$sh.log; head -2 $O/tl_$sh.log; grep -n "slice\|store\|epi" $O/tl_$sh.log | sed -n 1,60p
done
