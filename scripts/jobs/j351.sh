#!/bin/bash
# is the two-rank drop-in test stable on the committed code (without the neighbour-lane kernels of j349)?  8 runs; then base library 4 runs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j351; mkdir -p $O; cd $R
T="tests/test_dropin_scripts.py::test_sample_and_save_two_ranks_on_one_gpu"
p=0; for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest $T -q -x 2>&1 | grep -Eq "1 passed" && p=$((p+1)); done; echo "committed code: $p of 8 passed" | tee $O/two_ranks.log
p=0; for i in 1 2 3 4; do R2DM_HIP_LIB=$R/build_probe/lib_base.so timeout 600 python -m pytest $T -q -x 2>&1 | grep -Eq "1 passed" && p=$((p+1)); done; echo "base library: $p of 4 passed" | tee -a $O/two_ranks.log
