#!/bin/bash
# round 6: as j422 after out_conv got its round-5 loop back (the shared fp16 / fp32 loop body had compiled to 72 instead of 122 registers: 53 -> 117 us)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j423; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_fp16_mode.py -q -m gpu > $O/pytest_fp16.log 2>&1; tail -4 $O/pytest_fp16.log
JOB=j423 bash $R/scripts/jobs/j306.sh
