#!/bin/bash
# round 5: which unit?  The branch-free fir_up2 body's loads checked against a pattern (mode 5), its arithmetic evaluated twice and compared (mode 6) -- alone and next to the
# sharpest aggressor (a process looping the 1x1 convolution 128 -> 64 @ 64 x 1024)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j370; mkdir -p $O; cd $R
P=$R/build_probe/dpp_shift_probe
{ echo "== alone"; timeout 60 $P 4 5; timeout 60 $P 4 6
  echo "== next to the 1x1 convolution loop"
  rm -f /tmp/hog_ready; (SHAPE=128,64,64,1024,1,8 SECS=40 READY_FILE=/tmp/hog_ready timeout 90 python scripts/hog_conv_loop.py > /dev/null 2>&1 &)
  for i in $(seq 1 60); do [ -f /tmp/hog_ready ] && break; sleep 0.5; done
  timeout 60 $P 10 5; timeout 60 $P 10 6
  echo "== the rewritten kernel itself, same neighbour (control)"
  R2DM_HIP_LIB=$R/build_probe/lib_vb.so NEIGHBOUR=none SECS=6 timeout 100 python scripts/fir_up_soak.py 2>&1 | grep "^fir_up_soak"
} | tee $O/probe.log
