#!/bin/bash
# round 4: in-kernel timeline of the 128-channel-tile kernel (block 0), L2 128->128 and L3 256->256 at batch 8
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j201; mkdir -p $O
cd $R
for s in L2_128_128 L3_256_256; do
  for c in 128 64; do
    R2DM_F2_CO_TILE=$c B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_${s}_$c.log
    sed -n 1,3p $O/tl_${s}_$c.log
  done
done
