#!/bin/bash
# bisect of the rewritten fir_up2's two-rank failure: lib_va = the committed kernel with only the loop form changed (block-uniform trip count, clamped items);
# lib_vb = the committed loop with only the body changed (unconditional row loads + selects instead of the branch)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j361; mkdir -p $O; cd $R
for lib in lib_va lib_vb; do echo "== $lib"; R2DM_HIP_LIB=$R/build_probe/$lib.so REPS=5 timeout 900 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu | cut -c1-200; done | tee $O/diff.log
