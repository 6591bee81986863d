#!/bin/bash
# round 3: what do the fused GroupNorm statistics cost conv_f16x2?  Timing ablation with VALID data: the same input every forward; after the
# first forwards the f16x2 kernel stops writing its statistics slots, which therefore keep exactly the right numbers (arena addresses repeat)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j129; mkdir -p $O
cd $R
for rep in 1 2 3; do for d in -1 2000; do
R2DM_DEBUG_DROP_STATS=$d R2DM_HIP_LIB=$R/build_probe/lib_dropstats.so timeout 300 python scripts/time_forward_unchecked.py 2>&1 | grep -v amdgpu.ids | sed "s/^/drop statistics after $d launches: /"; done; done | tee $O/ab.log
