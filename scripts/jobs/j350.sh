#!/bin/bash
# bisect of the j349 failure (test_sample_and_save_two_ranks_on_one_gpu with the neighbour-lane kernels)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j350; mkdir -p $O; cd $R
T="tests/test_dropin_scripts.py::test_sample_and_save_two_ranks_on_one_gpu"
echo "== kernel tests"; timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "fir or out_conv" 2>&1 | grep -v amdgpu | tail -3
echo "== default"; timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== default again"; timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== R2DM_OUT_CONV=rows"; R2DM_OUT_CONV=rows timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== R2DM_FIR_STATS=0"; R2DM_FIR_STATS=0 timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== R2DM_FIR_NARROW=1 R2DM_FIR_STATS=0"; R2DM_FIR_NARROW=1 R2DM_FIR_STATS=0 timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== unet partition"; timeout 600 python -m pytest tests/test_hip_unet.py -q -m gpu -k "partition" 2>&1 | grep -v amdgpu | tail -2
