#!/bin/bash
# the two-rank drop-in test 30 times on the committed code (the neighbour-lane variants failed it 40 % of the time: is there a latent, rarer failure without them?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j357; mkdir -p $O; cd $R
T="tests/test_dropin_scripts.py::test_sample_and_save_two_ranks_on_one_gpu"
p=0; for i in $(seq 1 30); do timeout 600 python -m pytest $T -q -x 2>&1 | grep -Eq "1 passed" && p=$((p+1)); done; echo "committed code: $p of 30 passed" | tee $O/two_ranks.log
p=0; for i in $(seq 1 6); do timeout 600 python -m pytest tests/test_hip_configs.py -q -x -k "next_to_a_second or two_rank" 2>&1 | grep -Eq " passed" && p=$((p+1)); done; echo "neighbour / two-rank bench tests: $p of 6 runs passed" | tee -a $O/two_ranks.log
