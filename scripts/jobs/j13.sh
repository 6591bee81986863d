#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j13; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_configs.py -q -m gpu -s > $O/test_configs.log 2>&1; tail -30 $O/test_configs.log
timeout 900 python -m pytest tests/test_hip_unet.py -q -m gpu -s -k "golden or full_size or north_star" > $O/test_unet_s.log 2>&1; grep -v "^ *$" $O/test_unet_s.log | tail -60
