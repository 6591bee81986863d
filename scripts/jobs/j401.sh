#!/bin/bash
# round 6: in-kernel timelines (-DF2_PROF) of the level-1 64 -> 64 launch WITH a residual (the tile end the verdict names), the no-residual one, and the 128 -> 128 @ 32x512 one
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j401; mkdir -p $O
cd $R
for s in L1_64_64 L1_64_64_nores L2_128_128; do
  B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=900 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$s.log
  head -3 $O/tl_$s.log
done
