#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j37; mkdir -p $O
cd $R
timeout 600 python scripts/stress_shared_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/stress.log
