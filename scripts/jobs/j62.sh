#!/bin/bash
# wave_ops.h everywhere (conv_mfma / old epilogue statistics, GroupNorm kernels, attention, fir_up2): full suite + A/B vs lib_c1.so
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j62; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2 3; do for lib in build_probe/lib_c1.so r2dm_amd/libr2dm_hip.so; do
R2DM_HIP_LIB=$R/$lib timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench $lib', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"; done; done
