#!/bin/bash
# round 5: in-kernel timeline of the short decoder launches (one or two tiles per block) on the final code
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j332; mkdir -p $O; cd $R
for s in U3_128_128 U2_64_64 L4_256_256; do
  B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2prof.so MAXEV=700 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$s.log; sed -n 1,3p $O/tl_$s.log
done
SHAPES=U3_128_128,U2_64_64,L4_256_256,L4_512_512 ITERS=50 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_conv.log
