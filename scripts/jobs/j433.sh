#!/bin/bash
# tile-end instruction diet, three experiments and their sum: ds_write2 pairs in the turn (-DF2_TURN_W2), the range maximum from the sums of squares without branches (-DF2_RANGE_PQ),
# fp32 wave statistics (-DF2_STATS_F32).  Kernel parity with all three, then A/B on the step.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j433; mkdir -p $O
cd $R
R2DM_HIP_LIB=$R/build_probe/lib_all3.so timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_range.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests_all3.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for l in base6 w2 rpq statsf32 all3; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
