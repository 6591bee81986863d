#!/bin/bash
# round 4: 128-channel tile with two deferred quarters: probe, timeline, kernel + U-Net tests, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j203; mkdir -p $O
cd $R
SHAPES=L2_128_128,L3_256_256,L1_64_128,L2_128_256,L3_256_512 timeout 600 python scripts/wide_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_probe.log
for s in L2_128_128; do
  R2DM_F2_CO_TILE=128 B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_${s}_128.log
  grep -E "epi|==|tail" $O/tl_${s}_128.log | head -12
done
R2DM_F2_CO_TILE=128 timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "conv" 2>&1 | tail -5 | tee $O/pytest_kernels_wide.log
timeout 1200 python -m pytest tests/test_hip_unet.py tests/test_hip_kernels.py -q -x 2>&1 | tail -8 | tee $O/pytest_unet.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2; do
  for m in 64 auto; do
    if [ $m = 64 ]; then export R2DM_F2_CO_TILE=64; else unset R2DM_F2_CO_TILE; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tile=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
  done
done | tee $O/ab.log
