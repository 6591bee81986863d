#!/bin/bash
# round 3: where do the +1.5-2 % of the round-3 epilogue additions (observed maximum, weight scale) come from?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j80; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
b() { R2DM_HIP_LIB=$R/$1 timeout 300 env $2 python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline 2>$O/err.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('bench $1 $2', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))" || tail -3 $O/err.log; }
for rep in 1 2 3; do
b build_probe/bis_5bfe8bd/r2dm_amd/libr2dm_hip.so X=1
b r2dm_amd/libr2dm_hip.so X=1
b r2dm_amd/libr2dm_hip.so R2DM_NO_STAT_MAX=1
b r2dm_amd/libr2dm_hip.so "R2DM_NO_STAT_MAX=1 R2DM_NO_WSCALE=1"
b build_probe/lib_nostatmax.so R2DM_NO_STAT_MAX=1
done 2>&1 | tee $O/ab.log
