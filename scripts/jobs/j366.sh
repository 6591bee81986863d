#!/bin/bash
# what does the corruption of the branch-free fir_up2 next to a sampler process look like?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j366; mkdir -p $O; cd $R
R2DM_HIP_LIB=$R/build_probe/lib_vb.so NEIGHBOUR=sampler SECS=8 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep -v amdgpu | cut -c1-420 | tee $O/soak.log
