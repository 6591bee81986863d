#!/bin/bash
# kernel-level test of the FIR down-sampler's fused statistics + the two-mode U-Net test
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j342; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "fir" 2>&1 | grep -v amdgpu | tail -30
timeout 900 python -m pytest tests/test_hip_unet.py -q -m gpu -k "fir_down_statistics" 2>&1 | grep -v amdgpu | grep -E "assert|passed|failed|Error" | cut -c1-300 | tail -20
