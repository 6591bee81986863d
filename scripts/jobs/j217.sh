#!/bin/bash
# round 4: tile end of the 128-channel tile WITHOUT a residual (64 -> 128 @ 64x1024: 4 chunks per tile, 8 tiles per block)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j217; mkdir -p $O
cd $R
R2DM_F2_CO_TILE=128 B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=L1_64_128 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_L1_64_128.log
sed -n 1,3p $O/tl_L1_64_128.log; grep -E "epi|tail" $O/tl_L1_64_128.log | head -40
