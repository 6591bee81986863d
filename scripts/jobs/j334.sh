#!/bin/bash
# round 5: level-1 launches at batch 4 against batch 8 with the SAME tile (64 x 8 forced for Cout = 64, 128 x 4 for 64 -> 128):
# does a half batch (67 MB tensors: inside the 256 MB Infinity Cache) run in less than half the time?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j334; mkdir -p $O; cd $R
for rep in 1 2; do
for cfg in "64x8 L1_64_64,L1_128_64" "128 L1_64_128" "64 L1_64_64,L1_128_64"; do
  set -- $cfg
  for b in 8 4; do
    echo "== tile $1 batch $b"; B=$b R2DM_F2_CO_TILE=$1 SHAPES=$2 ITERS=40 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
  done
done
done | tee $O/bench_conv.log
