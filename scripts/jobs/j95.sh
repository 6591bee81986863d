#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j95; mkdir -p $O
cd $R; SEED=1 CASES=30 timeout 1500 python scripts/fuzz_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/fuzz1.log | cut -c1-260
