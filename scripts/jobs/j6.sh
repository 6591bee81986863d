#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j6; mkdir -p $O
cd $R
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_duo_rs.so; do
  echo "== $lib (duo forced)" >> $O/abl.log
  R2DM_HIP_LIB=$R/$lib R2DM_DUO_MIN=1 SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
done
cat $O/abl.log
cd $R/scripts
R2DM_HIP_LIB=$R/build_probe/lib_duo_rs_prof.so timeout 120 python duo_timeline.py > $O/timeline.log 2>&1
