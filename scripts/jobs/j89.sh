#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j89; mkdir -p $O
cd $R
{
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_sametile.so build_probe/lib_noloads.so; do
  echo "== $lib B=8"; R2DM_HIP_LIB=$R/$lib SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=50 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
for B in 1 2 4 8 16; do echo "== default lib B=$B"; B=$B SHAPES=L1_64_64,L1_64_128,L2_128_128 ITERS=50 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids; done
} | tee $O/ablation.log
cd /tmp
for B in 4 8 16; do python $R/bench.py --batch $B --steps 32 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('batch $B:', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'img/s', j['roofline']['board']['sclk_mhz'])"; done | tee $O/batch.log
