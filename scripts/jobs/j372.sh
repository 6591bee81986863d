#!/bin/bash
# round 5: the branch-free fir_up2 as is (lib_vb) | with 32 idle cycles between its last VALU write and its stores (lib_vh) | compiled WITHOUT packed-fp32 instructions
# (-fno-slp-vectorize: lib_vi), each next to the 1x1 convolution loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j372; mkdir -p $O; cd $R
for lib in lib_vb lib_vh lib_vi; do R2DM_HIP_LIB=$R/build_probe/$lib.so NEIGHBOUR=conv HOG_SHAPE=128,64,64,1024,1,8 SECS=8 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep "^fir_up_soak"; sleep 30; done | tee $O/soak.log
