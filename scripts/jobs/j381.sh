#!/bin/bash
# round 5, last session, final commit: final-sample validation at BASELINE configs[4]'s geometry (128x2048, 64 DDPM steps vs the CPU oracle on one noise tape) on the round-5 tiles,
# then the bitwise-repeatability soak (scripts/soak.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j381; mkdir -p $O; cd $R
RES=128x2048 STEPS=64 timeout 600 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | tee $O/validate_c4.log
timeout 420 python scripts/soak.py 2>&1 | grep -v amdgpu | tail -8 | tee $O/soak.log
