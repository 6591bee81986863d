#!/bin/bash
# round 5: minimal probe of the wave-wide DPP shift next to another process (scripts/probes/dpp_shift_probe.hip, built by the builder: build_probe/dpp_shift_probe):
# alone | next to a second instance of itself | next to a process looping a convolution of the library (out_conv; a 1x1 convolution with LDS) | next to a sampler
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j352; mkdir -p $O; cd $R
P=$R/build_probe/dpp_shift_probe
echo "== alone" | tee $O/probe.log; timeout 60 $P 8 | tee -a $O/probe.log
echo "== two instances" | tee -a $O/probe.log; (timeout 60 $P 12 > $O/second.log &) ; sleep 1; timeout 60 $P 8 | tee -a $O/probe.log; sleep 5; cat $O/second.log | tee -a $O/probe.log
for shape in 64,2,64,1024,3,8 512,512,8,128,1,8 64,64,64,1024,3,8; do
  echo "== next to hog_conv_loop SHAPE=$shape" | tee -a $O/probe.log
  rm -f /tmp/hog_ready; (SHAPE=$shape SECS=25 READY_FILE=/tmp/hog_ready timeout 60 python scripts/hog_conv_loop.py > /dev/null 2>&1 &)
  for i in $(seq 1 40); do [ -f /tmp/hog_ready ] && break; sleep 0.5; done
  timeout 60 $P 8 | tee -a $O/probe.log
  sleep 18
done
echo "== next to two ranks of the bulk sampler (the failing test's neighbour)" | tee -a $O/probe.log
(timeout 120 python -m pytest tests/test_dropin_scripts.py -q -k two_ranks > /dev/null 2>&1 &); sleep 4; timeout 60 $P 10 | tee -a $O/probe.log
