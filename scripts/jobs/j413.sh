#!/bin/bash
# round 6: pixels on the MFMA's row axis in the one-accumulator tiles (SWAP: no turn through LDS at the tile end) -- correctness, A/B against -DF2_NO_SWAP, timelines; then the narrow-presplit experiment (j412)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j413; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for l in build_probe/lib_noswap.so r2dm_amd/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$l timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
for s in L1_64_64 L1_64_64_nores L2_128_128; do
  B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=900 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$s.log
  head -1 $O/tl_$s.log; grep "epi begin\|slice end" $O/tl_$s.log | awk '{print $1}' | paste - - | awk '{print "tile end (epi begin -> E2)", $2 - $1}'
done | tee $O/tl_summary.txt
for i in 1 2 3; do
  for m in 0 1; do
    R2DM_F2_PRESPLIT_NARROW=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench narrow_presplit=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab_presplit.log
