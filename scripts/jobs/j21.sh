#!/bin/bash
# warp-specialised conv kernel: correctness + A/B against the pair/stream kernels in one job
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j21; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x > $O/pytest_kernels.log 2>&1; tail -8 $O/pytest_kernels.log
for spec in 1 0; do
  echo "== R2DM_SPEC=$spec"
  R2DM_SPEC=$spec SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128,L3_256_256,L4_512_512 ITERS=20 timeout 300 python scripts/bench_conv.py 2>&1 | grep -v amdgpu | tee $O/conv_spec$spec.log
done
for spec in 1 0 1 0; do
  R2DM_SPEC=$spec timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline > $O/bench_spec$spec.json 2> $O/bench_spec$spec.err
  python -c "
import json
j=json.load(open('$O/bench_spec$spec.json')); print('bench spec=$spec', j['value'], j['ms_per_step'], j['roofline'])
"
done
