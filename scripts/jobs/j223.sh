#!/bin/bash
# round 4: the plain-fmaf-chain yardstick (pieces = 5) next to the two-level fp32 kernel; step A/B of the one-accumulator tile up to Cin = 128 / 256
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j223; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_kernels.py -q -s -k "both_operand_splits" 2>&1 | grep -E "conv [0-9]|passed|failed|Error" | tee $O/splits.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2 3; do
  for m in 0 128 256; do
    R2DM_F2_WIDE_MAX_CIN=$m timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench wide_max_cin=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab.log
