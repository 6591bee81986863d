#!/bin/bash
# round 5: is it the SGPR-PAIR operand of the packed-fp32 instructions?  The branch-free fir_up2 as is (constants 0.25 / 0.75 as SGPR pairs: v_pk_mul_f32 v, v, s[34:35]) | with the
# two constants held in vector registers (all packed operands VGPR pairs: lib_vj), each next to the 1x1 convolution loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j373; mkdir -p $O; cd $R
for lib in lib_vb lib_vj; do R2DM_HIP_LIB=$R/build_probe/$lib.so NEIGHBOUR=conv HOG_SHAPE=128,64,64,1024,1,8 SECS=8 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep "^fir_up_soak"; sleep 30; done | tee $O/soak.log
