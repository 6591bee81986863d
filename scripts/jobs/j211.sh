#!/bin/bash
# round 4: the operand pre-pass in the fp16 bulk mode (one product per MAC: purely stager-bound) per layer shape; full-step A/B of the pre-pass in the parity mode
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j211; mkdir -p $O
cd $R
PIECES=1 timeout 600 python scripts/presplit_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/presplit_probe_fp16.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2 3; do
  for m in 0 256; do
    R2DM_F2_PRESPLIT_MIN_COUT=$m timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench presplit_min_cout=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
  done
done | tee $O/ab_presplit.log
for m in 0 128 256; do
    R2DM_F2_PRESPLIT_MIN_COUT=$m timeout 300 python $R/bench.py $A --precision fp16 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 presplit_min_cout=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
done | tee $O/ab_presplit_fp16.log
