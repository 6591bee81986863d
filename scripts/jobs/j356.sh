#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_hip_unet.py -q -m gpu -k "in_conv_channel or fir_down or noise_rows" 2>&1 | grep -v amdgpu | tail -3
