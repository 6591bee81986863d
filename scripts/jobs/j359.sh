#!/bin/bash
# round 5: which files differ, and by how much, when the neighbour-lane fir_up2 fails the two-rank scenario -- and does the SAME restructured kernel with loads instead of DPP (lib_nodpp) fail too?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j359; mkdir -p $O; cd $R
for lib in lib_dppup lib_nodpp; do echo "== $lib"; R2DM_HIP_LIB=$R/build_probe/$lib.so REPS=8 timeout 900 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu; done | tee $O/diff.log
