#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j45; mkdir -p $O
cd $R
{
for rep in 1 2; do
for lib in "$R/r2dm_amd/libr2dm_hip.so" "$R/build_probe/lib_stag2k.so" "$R/build_probe/lib_stag4k.so" "$R/build_probe/lib_stag8k.so"; do
  R2DM_HIP_LIB=$lib timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('lib=${lib##*/}', 'bench', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"
done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/stag.log
