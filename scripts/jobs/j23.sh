#!/bin/bash
# warp-specialised conv kernel: correctness, A/B, timeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j24; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x > $O/pytest_kernels.log 2>&1; tail -4 $O/pytest_kernels.log
S=L1_64_64,L1_128_64,L1_64_128,L2_128_128,L3_256_256,L4_512_512
for spec in 1 0; do
  echo "== R2DM_SPEC=$spec"
  R2DM_SPEC=$spec SHAPES=$S ITERS=20 timeout 300 python scripts/bench_conv.py 2>&1 | grep -v amdgpu | tee $O/conv_spec$spec.log
done
R2DM_HIP_LIB=$R/build_probe/lib_spec_prof.so SHAPES=L1_64_64,L3_256_256 timeout 300 python scripts/spec_timeline.py > $O/timeline.log 2>&1
head -150 $O/timeline.log
