#!/bin/bash
# GPU job 3: duo kernel ablations (timing only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j3; mkdir -p $O
cd $R
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_duo_NO_RAWLOAD.so build_probe/lib_duo_NO_XF.so build_probe/lib_duo_NO_EPI.so build_probe/lib_duo_all.so; do
  echo "== $lib (duo forced)" >> $O/abl.log
  R2DM_HIP_LIB=$R/$lib R2DM_DUO_MIN=1 SHAPES=L1_64_64,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
done
cat $O/abl.log
