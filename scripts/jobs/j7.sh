#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j7; mkdir -p $O
cd $R
for rep in 1 2; do
for lib in build_probe/lib_duo_rs.so build_probe/lib_duo_rs_pr.so; do
  echo "== $lib (duo forced)" >> $O/abl.log
  R2DM_HIP_LIB=$R/$lib R2DM_DUO_MIN=1 SHAPES=L1_64_64,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
done; done
cat $O/abl.log
