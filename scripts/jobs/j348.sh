#!/bin/bash
# round 5: out_conv with the neighbour pixels from the neighbour lanes (conv_direct_dpp_kernel) -- bit-identity tests, then the launch timed in its variants
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j348; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "out_conv or conv3x3_ring" 2>&1 | grep -v amdgpu | tail -4
for v in rows dpp24 dpp28 dpp14 dpp18 rows dpp24 dpp28 dpp14 dpp18; do
  echo -n "$v: "; R2DM_OUT_CONV=$v SHAPES=L1_64_2 ITERS=50 python scripts/bench_conv.py 2>&1 | grep -v amdgpu | tail -1
done | tee $O/variants.log
for v in rows dpp24 dpp14; do echo -n "B=32 $v: "; B=32 R2DM_OUT_CONV=$v SHAPES=L1_64_2 ITERS=20 python scripts/bench_conv.py 2>&1 | grep -v amdgpu | tail -1; done | tee -a $O/variants.log
