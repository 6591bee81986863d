#!/bin/bash
# round 5: is it the STORE's data registers being overwritten too early (lanes 48-63 = the last pass of the store-data read)?  The branch-free fir_up2 with
# s_nop 7 x 4 behind its two 16-byte stores (lib_vf) | with s_waitcnt vmcnt(0) behind them (lib_vg) | as is (lib_vb), each next to the 1x1 convolution loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j371; mkdir -p $O; cd $R
for lib in lib_vb lib_vf lib_vg; do R2DM_HIP_LIB=$R/build_probe/$lib.so NEIGHBOUR=conv HOG_SHAPE=128,64,64,1024,1,8 SECS=8 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep "^fir_up_soak"; sleep 30; done | tee $O/soak.log
