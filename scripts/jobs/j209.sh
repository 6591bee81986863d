#!/bin/bash
# round 4: the operand pre-pass (presplit.hip) per layer shape: bit-equality + timing
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j209; mkdir -p $O
cd $R
timeout 600 python scripts/presplit_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/presplit_probe.log
