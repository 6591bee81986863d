#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j27; mkdir -p $O
cd $R
{
for w in L1_64_64 L1_64_128 L2_128_128 L3_256_256; do
  PIECES=2 R2DM_SPEC=0 WORK=$w SECS=2 timeout 60 python scripts/power_probe.py
  PIECES=2 R2DM_SPEC=1 WORK=$w SECS=2 timeout 60 python scripts/power_probe.py
done
} 2>&1 | grep -v amdgpu.ids | tee $O/power.log
