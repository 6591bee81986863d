#!/bin/bash
# round 5: in-kernel timelines (-DF2_PROF) of the 64 x 8 and 64 x 4 tiles on the level-1 64 -> 64 layer, with and without a residual
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j302; mkdir -p $O
cd $R
for t in 64x8 64; do for s in L1_64_64 L1_64_64_nores; do
  R2DM_F2_CO_TILE=$t B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=700 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_${t}_$s.log
  head -3 $O/tl_${t}_$s.log
done; done
