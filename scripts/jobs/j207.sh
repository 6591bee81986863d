#!/bin/bash
# round 4: wide tile with the residual tile touched a chunk ahead: timeline, probe, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j207; mkdir -p $O
cd $R
R2DM_F2_CO_TILE=128 B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=L2_128_128 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_L2_128.log
grep -E "epi|==|tail" $O/tl_L2_128.log | head -24
SHAPES=L2_128_128,L3_256_256,L1_64_128 timeout 600 python scripts/wide_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_probe.log
timeout 600 python -m pytest tests/test_hip_kernels.py -q -s -k "both_operand_splits" 2>&1 | grep -E "conv [0-9]|passed|failed|Error" | tee $O/splits.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2; do
  for m in 64 auto; do
    if [ $m = 64 ]; then export R2DM_F2_CO_TILE=64; else unset R2DM_F2_CO_TILE; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tile=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
  done
done | tee $O/ab.log
