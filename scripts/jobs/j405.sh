#!/bin/bash
# round 6: does de-phasing shorten a tile end?  In-kernel timeline of block 0 (never delayed) of the level-1 64 -> 64 launches with half of the blocks started 0 / 12 k / 24 k cycles late (R2DM_F2_STAGGER)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j405; mkdir -p $O
cd $R
for st in 0 12000 24000; do for s in L1_64_64 L1_64_64_nores; do
  R2DM_F2_STAGGER=$st B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=900 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_${s}_st$st.log
  echo "== stagger $st $s"; head -1 $O/tl_${s}_st$st.log; grep "epi begin\|epi end" $O/tl_${s}_st$st.log | awk '{print $1, $5, $6}' | paste - - | awk '{print "tile end", $4 - $1}'
done; done | tee $O/summary.txt
