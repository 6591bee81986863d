#!/bin/bash
# round 3: conv_f16x2 with the pixel loads spread through the transform (three register sets): correctness, then timing A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j90; mkdir -p $O
cd $R
R2DM_HIP_LIB=$R/build_probe/lib_spread.so timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "conv3x3 or conv_fused or conv_golden" 2>&1 | tail -3
R2DM_HIP_LIB=$R/build_probe/lib_spread.so timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_fp16_mode.py -q -m gpu -x 2>&1 | tail -3
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_spread.so; do
  echo "== $lib"; R2DM_HIP_LIB=$R/$lib SHAPES=L1_64_64,L1_64_128,L2_128_128,L3_256_256,L4_512_512,L4_256_256 ITERS=50 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
done | tee $O/conv.log
cd /tmp
for rep in 1 2 3; do for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_spread.so; do
R2DM_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('bench $lib', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; done; done 2>&1 | tee $O/ab.log
