#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j9; mkdir -p $O
cd $R
R2DM_HIP_LIB=$R/build_probe/lib_duo_rs.so R2DM_DUO_MIN=1 timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv or group_norm" > $O/test.log 2>&1; tail -3 $O/test.log
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_duo_rs.so; do
  echo "== $lib (duo forced)" >> $O/abl.log
  R2DM_HIP_LIB=$R/$lib R2DM_DUO_MIN=1 SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
done
echo "== pair" >> $O/abl.log
R2DM_DUO_MIN=100000000 SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
cat $O/abl.log
cd $R/scripts
R2DM_HIP_LIB=$R/build_probe/lib_duo_rs_prof.so timeout 120 python duo_timeline.py > $O/timeline_rs.log 2>&1
