#!/bin/bash
# round 5, first GPU job: the 64 x 8 one-accumulator tile (conv_f16x2.hip, Geo<2, 4>) -- kernel tests, whole-net tests, step A/B against R2DM_F2_TALL=0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j300; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "conv3x3" > $O/pytest_conv.log 2>&1; tail -15 $O/pytest_conv.log
timeout 900 python -m pytest tests/test_hip_unet.py -q -x > $O/pytest_unet.log 2>&1; tail -5 $O/pytest_unet.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in 0 1; do
    R2DM_F2_TALL=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tall=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_tall.log
for m in 64 64x8; do
  echo "== R2DM_F2_CO_TILE=$m"; R2DM_F2_CO_TILE=$m SHAPES=L1_64_64,L1_128_64 ITERS=20 timeout 300 python scripts/bench_conv.py 2>&1 | grep -v amdgpu
done | tee $O/shapes.log
