#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j318; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_fp16_mode.py -q -k "storage" 2>&1 | tail -25 | cut -c1-200
