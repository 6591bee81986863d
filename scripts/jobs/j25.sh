#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j25; mkdir -p $O
cd $R
R2DM_HIP_LIB=$R/build_probe/lib_spec_prof.so MAXEV=0 SHAPES=L1_64_64,L1_128_64,L2_128_128 timeout 300 python scripts/spec_timeline.py 2>&1 | grep -v amdgpu | tee $O/clock.log
rocm-smi --showclocks 2>&1 | head -30
