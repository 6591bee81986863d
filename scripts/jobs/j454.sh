#!/bin/bash
# statistics arguments pinned as opaque scalars (no re-loads from the argument segment in the tile ends): parity + A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j454; mkdir -p $O
cd $R
R2DM_HIP_LIB=$R/build_probe/lib_pin.so timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for l in fin pin; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
