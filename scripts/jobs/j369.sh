#!/bin/bash
# final code next to the sharpest aggressor found this session (a process looping the level-1 1x1 convolution 128 -> 64: LDS-holding, shares CUs): forwards of every precision mode
# compared with the one computed alone (scripts/coresidency_probe.py), 3 x 1000 forwards, twice; then the drop-in scenario (scripts/two_rank_diff.py, 6 repetitions)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j369; mkdir -p $O; cd $R
for i in 1 2; do MODE=process HOG_SHAPE=128,64,64,1024,1,8 REPS=1000 timeout 300 python scripts/coresidency_probe.py 2>&1 | grep coresidency_probe; done | tee $O/soak.log
REPS=6 timeout 600 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu | tail -2 | tee -a $O/soak.log
