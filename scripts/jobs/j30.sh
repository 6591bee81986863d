#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j30; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log; grep -a "R2DMError:" $O/pytest_gpu.log | head -3
