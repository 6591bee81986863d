#!/bin/bash
# round 4: timeline of the fp16 bulk mode (one product per MAC), L3 256->256, four vs eight staging waves
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j220; mkdir -p $O
cd $R
for v in f2_prof4 f2_prof; do
  PIECES=1 R2DM_F2_CO_TILE=64 B=8 R2DM_HIP_LIB=$R/build_probe/lib_$v.so MAXEV=400 SHAPES=L3_256_256 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$v.log
  echo "== $v"; sed -n 1,1p $O/tl_$v.log; sed -n 70,100p $O/tl_$v.log
done
