#!/bin/bash
# round 3: (a) what the old out_conv kernel computes wrongly next to a neighbour (diag); (b) first GPU run of the round-3 code:
# observed-max range guard, per-layer weight scale, fp16 one-product mode, new tests; (c) A/B of the step time against round 2's library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j78; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
(cd build_probe/bis_d5e0cd1 && HOG_CMD="cd $R && SHAPE=512,512,8,128,1,8 SECS=200 python $R/scripts/hog_conv_loop.py" timeout 400 python $R/scripts/diag_old_direct.py 2>&1 | grep -v amdgpu.ids | tee $O/diag.log)
timeout 1500 python -m pytest tests/test_hip_range.py tests/test_hip_fp16_mode.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $O/pytest_new.log; tail -40 $O/pytest_new.log | cut -c1-300
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids > $O/pytest_old.log; tail -15 $O/pytest_old.log | cut -c1-300
for rep in 1 2; do for lib in build_probe/bis_5bfe8bd/r2dm_amd/libr2dm_hip.so r2dm_amd/libr2dm_hip.so; do
R2DM_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('bench $lib', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3), j['config']['clock_prewarm'])"; done; done 2>&1 | tee $O/ab.log
