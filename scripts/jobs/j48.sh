#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j48; mkdir -p $O
cd $R
R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=400 SHAPES=L1_128_64,L3_256_256 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/timeline.log
grep -n "==\|epi" $O/timeline.log | head -30
