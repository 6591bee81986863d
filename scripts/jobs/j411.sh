#!/bin/bash
# round 6: in-kernel timelines of the small up-path launches (one tile per CU) and the 512 -> 512 layer: where do 35-43 us go?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j411; mkdir -p $O
cd $R
for s in L4_256_256 U3_128_128 U2_64_64 L4_512_512; do
  B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=900 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$s.log
  head -2 $O/tl_$s.log
done
