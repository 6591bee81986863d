#!/bin/bash
# bisect of the j340 failure (test_sample_and_save_matches_api): which of the session's small-kernel changes breaks it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j341; mkdir -p $O; cd $R
T="tests/test_dropin_scripts.py::test_sample_and_save_matches_api"
echo "== default"; timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== R2DM_FIR_STATS=0"; R2DM_FIR_STATS=0 timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== R2DM_FEW_IN_SPLIT=1"; R2DM_FEW_IN_SPLIT=1 timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== both"; R2DM_FIR_STATS=0 R2DM_FEW_IN_SPLIT=1 timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== base lib"; R2DM_HIP_LIB=$R/build_probe/lib_base.so timeout 600 python -m pytest $T -q -x 2>&1 | grep -v amdgpu | tail -2
echo "== whole suite"; timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | tail -15
