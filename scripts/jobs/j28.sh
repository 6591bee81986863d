#!/bin/bash
# f16x2 convolution: kernel parity, U-Net parity, speed vs bf16x3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j28; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -s > $O/pytest_kernels.log 2>&1; grep "^conv \|passed\|failed\|Error\|error" $O/pytest_kernels.log | tail -40
{
for w in L1_64_64 L1_128_64 L2_128_128 L3_256_256 L4_512_512; do
  PIECES=2 WORK=$w SECS=2 timeout 60 python scripts/power_probe.py
  PIECES=3 WORK=$w SECS=2 timeout 60 python scripts/power_probe.py
done
} 2>&1 | grep -v amdgpu.ids | tee $O/power.log
timeout 1200 python -m pytest tests/test_hip_unet.py tests/test_hip_configs.py -q -m gpu -x -s > $O/pytest_unet.log 2>&1; grep -v "^$" $O/pytest_unet.log | tail -40
