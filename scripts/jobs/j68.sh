#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j68; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2 3; do for lib in ${LIBS:-build_probe/lib_c4.so r2dm_amd/libr2dm_hip.so}; do
R2DM_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 128 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('bench $lib', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3), [round(e['ms_per_step'],3) for e in r['other_conv_kernels']])"; done; done
