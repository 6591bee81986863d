#!/bin/bash
# GPU job 2: duo kernel first light -- correctness (forced for every shallow shape) then timing vs the pair kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j2; mkdir -p $O
cd $R
R2DM_DUO_MIN=1 timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv" > $O/test_duo_forced.log 2>&1; echo "rc=$?" >> $O/test_duo_forced.log
tail -15 $O/test_duo_forced.log
for v in 1 1000000; do
  echo "== R2DM_DUO_MIN=$v" >> $O/bench_conv.log
  R2DM_DUO_MIN=$v SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py >> $O/bench_conv.log 2>&1
done
cat $O/bench_conv.log
