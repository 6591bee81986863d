#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j32; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x > $O/pytest_kernels.log 2>&1; tail -2 $O/pytest_kernels.log
R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=70 SHAPES=L1_64_64 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/timeline.log
head -75 $O/timeline.log
{
for w in L1_64_64 L1_128_64 L2_128_128 L3_256_256 L4_512_512; do
  PIECES=2 WORK=$w SECS=2 timeout 60 python scripts/power_probe.py
done
} 2>&1 | grep -v amdgpu.ids | tee $O/power.log
