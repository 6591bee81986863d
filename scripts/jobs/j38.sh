#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j38; mkdir -p $O
cd $R
{
timeout 300 python scripts/stress_shared_forward.py
PRECISION=fp32-bf16x3 timeout 300 python scripts/stress_shared_forward.py
R2DM_CONV_ALGO=f32 timeout 300 python scripts/stress_shared_forward.py
} 2>&1 | grep -v amdgpu.ids | tee $O/stress.log
