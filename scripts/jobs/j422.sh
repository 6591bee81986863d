#!/bin/bash
# round 6: fp16-mode tests + whole GPU suite on the tree with fp16 storage of the full-resolution levels, then the round's evidence set (r06a: scripts/jobs/j306.sh)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j422; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_fp16_mode.py -q -m gpu > $O/pytest_fp16.log 2>&1; tail -4 $O/pytest_fp16.log
JOB=j422 bash $R/scripts/jobs/j306.sh
