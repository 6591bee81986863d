#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j34; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_configs.py -q -m gpu -x -k "range" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
