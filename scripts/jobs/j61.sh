#!/bin/bash
# generic A/B: kernel + unet tests on the in-tree library, epilogue timeline, bench of $LIBS (default: lib_c1 = previous commit, in-tree)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j61; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=2000 SHAPES=${TL_SHAPE:-L1_64_64} timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl.log
sed -n 1,1p $O/tl.log; grep "  M  M epi" $O/tl.log | sed -n 12,23p
for rep in 1 2 3; do for lib in ${LIBS:-build_probe/lib_c1.so r2dm_amd/libr2dm_hip.so}; do
R2DM_HIP_LIB=$R/$lib timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench $lib', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"; done; done
