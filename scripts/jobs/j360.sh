#!/bin/bash
# round 5: the two-rank scenario at 64 x 1024 (scripts/two_rank_diff.py) on the COMMITTED library and on the session's base library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j360; mkdir -p $O; cd $R
echo "== committed"; REPS=6 timeout 900 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu | cut -c1-400 | tee $O/diff.log
echo "== base"; R2DM_HIP_LIB=$R/build_probe/lib_base.so REPS=4 timeout 900 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu | cut -c1-400 | tee -a $O/diff.log
