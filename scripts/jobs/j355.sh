#!/bin/bash
# round 5: folded GroupNorm with all four slot rounds of a level-1 group requested at once (-DF2_FOLD4; 8-11 spilled registers in the 64 x 8 tile) -- bit-identity, then A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j355; mkdir -p $O; cd $R
R2DM_HIP_LIB=$R/build_probe/lib_fold4.so timeout 600 python -m pytest tests/test_hip_unet.py -q -m gpu -k "folded_into_its_consumer" 2>&1 | grep -v amdgpu | tail -2
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4; do
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line default
  R2DM_HIP_LIB=$R/build_probe/lib_fold4.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line fold4
done | tee $O/ab.log
