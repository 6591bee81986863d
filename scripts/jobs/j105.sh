#!/bin/bash
# round 3: conv_f16x2 launch overhead (prologue order, tail residual prefetch): tests, timeline, A/B against the previous library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j105; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_range.py tests/test_hip_fp16_mode.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.log
for sh in U3_128_128 L1_64_64; do
B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=2000 SHAPES=$sh timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$sh.log; head -14 $O/tl_$sh.log; tail -7 $O/tl_$sh.log
done
cd /tmp
for rep in 1 2 3; do for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so; do
R2DM_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; done; done 2>&1 | tee $O/ab.log
