#!/bin/bash
# round 3, weak #1: bisect, second step (j75: d5e0cd1 fails, 87d5e9d is clean): the two commits in between, and the bf16x3 split on both sides
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j76; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
{
for c in 9756b32 1c7a0dc; do
  echo "== tree of $c"
  (cd build_probe/bis_$c && timeout 300 python scripts/stress_shared_forward.py 2>&1 | grep -v amdgpu.ids | grep forward)
done
for c in d5e0cd1 1c7a0dc 87d5e9d; do
  echo "== tree of $c, PRECISION=fp32-bf16x3"
  (cd build_probe/bis_$c && PRECISION=fp32-bf16x3 ITERS=300 timeout 300 python scripts/stress_shared_forward.py 2>&1 | grep -v amdgpu.ids | grep forward)
done
} 2>&1 | tee $O/bisect.log
