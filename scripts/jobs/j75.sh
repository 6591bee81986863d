#!/bin/bash
# round 3, weak #1: bisect.  Same box, one job: the tree of 9e66794 fails (j74: 150 and 88 of 150 forwards), HEAD does not.
# Which commit in between fixed it?  Every tree runs its own copy of scripts/stress_shared_forward.py (as committed then).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j75; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
for c in ${COMMITS:-43d0cd6 652b236 39ddd41 d5e0cd1 87d5e9d 3f214d4 8c05182}; do
  echo "== tree of $c"
  (cd build_probe/bis_$c && timeout 300 python scripts/stress_shared_forward.py 2>&1 | grep -v amdgpu.ids | grep forward)
done 2>&1 | tee $O/bisect.log
