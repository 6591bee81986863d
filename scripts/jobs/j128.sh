#!/bin/bash
# round 3, final code (100effb+): final-sample validation against the oracle on an identical noise tape: 256-step DDPM, 32-step DDIM, 128x2048
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j128; mkdir -p $O
cd $R
{ timeout 1500 python scripts/validate_256.py; MODE=ddim STEPS=32 timeout 600 python scripts/validate_256.py; RES=128x2048 STEPS=32 timeout 1500 python scripts/validate_256.py; PRECISION=fp16 timeout 1500 python scripts/validate_256.py; } 2>&1 | grep -v amdgpu.ids | tee $O/validate.log
