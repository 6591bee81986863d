#!/bin/bash
# round 4: 128-channel tile, one deferred slice + epilogue diet: probe, timeline, tests, 256-step validation with every Cout % 128 layer on the wide tile, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j205; mkdir -p $O
cd $R
SHAPES=L2_128_128,L3_256_256,L1_64_128,L1_64_64 timeout 600 python scripts/wide_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_probe.log
R2DM_F2_CO_TILE=128 B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=1000 SHAPES=L2_128_128 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_L2_128.log
grep -E "epi|==|tail" $O/tl_L2_128.log | head -12
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -x 2>&1 | tail -4 | tee $O/pytest.log
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -s -k "both_operand_splits" 2>&1 | grep "^conv" | tee $O/splits.log
for t in 128 64; do R2DM_F2_CO_TILE=$t timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/tile=$t: /"; done | tee $O/validate_256.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2; do
  for m in 64 auto; do
    if [ $m = 64 ]; then export R2DM_F2_CO_TILE=64; else unset R2DM_F2_CO_TILE; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tile=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
  done
done | tee $O/ab.log
