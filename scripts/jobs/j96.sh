#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j96; mkdir -p $O
cd $R; timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "attention" -s 2>&1 | grep -v amdgpu | tail -25 | cut -c1-200
SEED=2 CASES=40 timeout 1500 python scripts/fuzz_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/fuzz2.log | grep -E "FAIL|rejected|cases," | cut -c1-260
