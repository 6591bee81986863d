#!/bin/bash
# round 6: fp16 storage of every activation of the two full-resolution levels in the one-plane mode (R2DM_FP16_STORAGE=2, default) -- tests, then bench --precision fp16 against storage level 1 (round 5) and 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j421; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_fp16_mode.py tests/test_hip_kernels.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --precision fp16"
for i in 1 2 3; do
  for m in 1 2; do
    R2DM_FP16_STORAGE=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 storage=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
R2DM_FP16_STORAGE=0 timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 storage=0', round(j['ms_per_step'],3), round(j['value'],3))" | tee -a $O/ab.log
