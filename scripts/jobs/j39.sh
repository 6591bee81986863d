#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j39; mkdir -p $O
cd $R
timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu.ids | tee $O/validate.log
