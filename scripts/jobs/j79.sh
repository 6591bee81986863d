#!/bin/bash
# round 3: (a) MFMA(f16) next to VALU of a second wave: ArchVGPR vs AccVGPR accumulators (calibration probe);
# (b) where the +1.2 % of the round-3 epilogue additions come from; issue-priority variants of conv_f16x2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j79; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
timeout 200 $R/build_probe/mfma_f16_valu_overlap 2>&1 | tee $O/overlap.log
b() { R2DM_HIP_LIB=$R/$1 timeout 300 env $2 python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('bench $1 $2', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; }
for rep in 1 2; do
b build_probe/bis_5bfe8bd/r2dm_amd/libr2dm_hip.so X=1
b r2dm_amd/libr2dm_hip.so X=1
b r2dm_amd/libr2dm_hip.so R2DM_NO_STAT_MAX=1
b r2dm_amd/libr2dm_hip.so "R2DM_NO_STAT_MAX=1 R2DM_NO_WSCALE=1"
b build_probe/lib_prio_s3.so X=1
b build_probe/lib_prio_m3.so X=1
b build_probe/lib_prio_s1m0.so X=1
done 2>&1 | tee $O/ab.log
