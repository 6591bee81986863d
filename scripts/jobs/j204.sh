#!/bin/bash
# round 4: is the wide tile's end bound by the memory burst of 256 blocks in lockstep or by its own instruction stream?  timeline at batch 1 / 2 / 8
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j204; mkdir -p $O
cd $R
for b in 1 2 8; do
  R2DM_F2_CO_TILE=128 B=$b R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=L2_128_128 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_b$b.log
  echo "== batch $b"; grep -E "epi|==|tail" $O/tl_b$b.log | head -8
done
