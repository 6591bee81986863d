#!/bin/bash
# round 5: de-phasing experiment -- every other block of an XCD starts R2DM_F2_STAGGER clock ticks late (level 1 is HBM-bound in lockstep bursts at the tile ends)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j307; mkdir -p $O
cd $R
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2; do
  for m in 0 8000 16000 28000; do
    R2DM_F2_STAGGER=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench stagger=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_stagger.log
for m in 0 16000; do
  echo "== R2DM_F2_STAGGER=$m"; R2DM_F2_STAGGER=$m SHAPES=L1_64_64,L1_128_64,L2_128_128 ITERS=20 timeout 300 python scripts/bench_conv.py 2>&1 | grep -v amdgpu
done | tee $O/shapes.log
