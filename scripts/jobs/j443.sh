#!/bin/bash
# gn_moments with exact power-of-two scaling instead of two fp64 divisions: bit-identity against the previous library, then A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j443; mkdir -p $O
cd $R
OUT=/tmp/ab.pt R2DM_HIP_LIB=$R/build_probe/lib_head2.so timeout 300 python scripts/ab_bits.py 2>&1 | grep ab_bits | tee $O/bits.log
OUT=/tmp/ab.pt R2DM_HIP_LIB=$R/build_probe/lib_p2.so timeout 300 python scripts/ab_bits.py 2>&1 | grep ab_bits | tee -a $O/bits.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for l in head2 p2; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
