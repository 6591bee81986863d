#!/bin/bash
# experiment: GroupNorm statistics of the conv_f16x2 tile ends reduced in fp32 over the wave (256 pixels per channel block), fp64 from the slot on (-DF2_STATS_F32): A/B on the step,
# and what it does to the parity figures (north-star final sample, full-size forward vs fp64)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j432; mkdir -p $O
cd $R
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for l in base6 statsf32; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
for l in base6 statsf32; do
  echo "== $l"
  R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 900 python -m pytest tests/test_hip_unet.py -m gpu -x -q -s -k "north_star or full_size_vs_oracle or batch8_full_size" 2>&1 | grep -i "max\|passed\|failed" | tail -8
done | tee $O/parity.log
