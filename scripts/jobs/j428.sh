#!/bin/bash
# prologue order (chunks 1, 2 requested behind the transform of chunk 0): parity of the conv tests, then bench alternating old order (lib_prov1.so) and new
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j428; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for l in build_probe/lib_prov1.so r2dm_amd/csrc/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$l timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
