#!/bin/bash
# round 4: timeline of L4 512->512 with the stagers' transform vs the pre-pass + DMA-only stagers
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j210; mkdir -p $O
cd $R
for pre in 0 1; do
  R2DM_F2_PRESPLIT=$pre R2DM_F2_CO_TILE=64 B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=400 SHAPES=L4_512_512 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_L4_pre$pre.log
  echo "== presplit $pre"; sed -n 1,3p $O/tl_L4_pre$pre.log; sed -n 60,110p $O/tl_L4_pre$pre.log
done
