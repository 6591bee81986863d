#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j88; mkdir -p $O
cd $R
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_noloads.so build_probe/lib_nodma.so build_probe/lib_noloads_nodma.so build_probe/lib_noloads_nodma_noxf.so build_probe/lib_noxf.so; do
  echo "== $lib"; R2DM_HIP_LIB=$R/$lib SHAPES=L1_64_64,L2_128_128,L4_512_512,L1_64_128 ITERS=50 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
done | tee $O/ablation.log
