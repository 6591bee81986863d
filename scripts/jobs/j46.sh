#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j46; mkdir -p $O
cd $R
{
for rep in 1 2; do
for mt in 128 256; do
  R2DM_F2_MIN_TILES=$mt timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('min_tiles=$mt', 'bench', round(j['value'],3), round(j['ms_per_step'],3), [ (e['kernel'][:14], e['launches_per_step'], round(e['ms_per_step'],3)) for e in [j['roofline']['dominant_kernel']]+j['roofline']['other_conv_kernels']])"
done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/mt.log
