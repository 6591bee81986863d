#!/bin/bash
# A/B (alternating, 128 steps): $LIBS; first the kernel + unet tests on the in-tree library and a timeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j64; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=400 SHAPES=L1_64_64 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl.log
sed -n 1,1p $O/tl.log; sed -n 100,130p $O/tl.log
for rep in 1 2 3; do for lib in ${LIBS:-build_probe/lib_c2.so r2dm_amd/libr2dm_hip.so}; do
R2DM_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 128 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('bench $lib', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; done; done
