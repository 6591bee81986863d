#!/bin/bash
# round 3: three-way A/B: 1ee35c3 (base) | 94115ae (prologue + tail residuals) | working tree (last tile split with the stagers)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j107; mkdir -p $O
cd /tmp
for rep in 1 2 3; do for lib in build_probe/lib_nosplit.so r2dm_amd/libr2dm_hip.so build_probe/lib_prev.so; do
R2DM_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; done; done 2>&1 | tee $O/ab.log
