#!/bin/bash
# three-way split of the stagers' transform: tests, timeline, A/B against the committed kernel (build_probe/lib_base.so) on one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j54; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=90 SHAPES=L1_64_64 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/timeline.log
sed -n 1,2p $O/timeline.log; sed -n 30,92p $O/timeline.log
for rep in 1 2; do for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so $EXTRA_LIBS; do
R2DM_HIP_LIB=$R/$lib timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench $lib', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"; done; done
