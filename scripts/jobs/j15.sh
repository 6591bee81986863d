#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j15; mkdir -p $O
cd $R
python scripts/check_workspace_independence.py > $O/ws.log 2>&1; cat $O/ws.log | grep -v amdgpu
RES=16,128 python scripts/check_workspace_independence.py 2>&1 | grep -v amdgpu
