#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j11; mkdir -p $O
cd $R
R2DM_DUO_MIN=1 timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv or group_norm" > $O/test.log 2>&1; tail -2 $O/test.log
for v in 1 100000000; do
echo "== DUO_MIN=$v" >> $O/abl.log
R2DM_DUO_MIN=$v SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
done
cat $O/abl.log
cd $R/scripts
R2DM_HIP_LIB=$R/build_probe/lib_duo_prof.so timeout 120 python duo_timeline.py > $O/timeline.log 2>&1
