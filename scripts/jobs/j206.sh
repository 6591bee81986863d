#!/bin/bash
# round 4: finer stamps inside the wide tile's end; the operand-split test's numbers; new robustness tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j206; mkdir -p $O
cd $R
R2DM_F2_CO_TILE=128 B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=L2_128_128 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_L2_128.log
grep -E "epi|==|tail" $O/tl_L2_128.log | head -24
timeout 600 python -m pytest tests/test_hip_kernels.py -q -s -k "both_operand_splits" 2>&1 | grep -E "^conv|passed|failed" | tee $O/splits.log
timeout 900 python -m pytest tests/test_hip_range.py tests/test_dropin_scripts.py tests/test_hip_kernels.py -q -k "range or sample_and_save or lidar" 2>&1 | tail -15 | tee $O/robust.log
