#!/bin/bash
# round 4: more final-sample validation on the round's code: 128x2048 (BASELINE configs[4] geometry), the fp16 bulk mode, the bf16x3 mode
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j218; mkdir -p $O
cd $R
(RES=128x2048 STEPS=64 timeout 1500 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1
 PRECISION=fp32-bf16x3 STEPS=128 timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/^/precision fp32-bf16x3: /'
 PRECISION=fp16 STEPS=128 timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/^/precision fp16 (reduced bulk mode, own tolerance class): /') | tee $O/validate_more.log
