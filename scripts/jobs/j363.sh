#!/bin/bash
# is the rewritten fir_up2's two-rank failure a matter of TIMING?  The committed kernel with a pure delay at its start (s_sleep 127 x 4 | x 40 per wave)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j363; mkdir -p $O; cd $R
for lib in lib_vc lib_ve; do echo "== $lib"; R2DM_HIP_LIB=$R/build_probe/$lib.so REPS=5 timeout 900 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu | cut -c1-160; done | tee $O/diff.log
