#!/bin/bash
# round 3, weak #1 (continued): j73 did not reproduce the round-2 failure with HEAD.  Same box, same job: the tree of commit
# 9e66794 (the one that measured 96 of 150; build_probe/old38) with its own script, then HEAD (B = 2 and 8, both splits),
# then the two-rank bench test on the default kernels.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j74; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
{
echo "== tree of 9e66794 (round 2, j38), its own script"
(cd build_probe/old38 && timeout 300 python scripts/stress_shared_forward.py; PRECISION=fp32-bf16x3 timeout 300 python scripts/stress_shared_forward.py)
echo "== HEAD"
TAG=head timeout 300 python scripts/stress_shared_forward.py
TAG=head ITERS=400 timeout 300 python scripts/stress_shared_forward.py
TAG=head B=8 timeout 400 python scripts/stress_shared_forward.py
TAG=head PRECISION=fp32-bf16x3 timeout 300 python scripts/stress_shared_forward.py
echo "== tree of 9e66794 again"
(cd build_probe/old38 && timeout 300 python scripts/stress_shared_forward.py)
} 2>&1 | grep -v amdgpu.ids | tee $O/stress_forward.log
timeout 900 python -m pytest tests/test_hip_configs.py -q -m gpu -x -k "two_rank" 2>&1 | tail -5 | tee $O/two_rank.log
