#!/bin/bash
# round 5: the DPP probe in the FIR variants' exact shape (mode 2: the DPP result overwritten in seam lanes by a load under a partial EXEC mask), alone and next to
# the two-rank bulk sampler / a second instance / a convolution loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j358; mkdir -p $O; cd $R
P=$R/build_probe/dpp_shift_probe
echo "== alone" | tee $O/probe.log; timeout 60 $P 6 2 | tee -a $O/probe.log
echo "== two instances" | tee -a $O/probe.log; (timeout 60 $P 10 2 > $O/second.log &) ; sleep 1; timeout 60 $P 6 2 | tee -a $O/probe.log; sleep 5; cat $O/second.log | tee -a $O/probe.log
for k in 1 2 3; do
echo "== next to the two-rank bulk sampler ($k)" | tee -a $O/probe.log
(timeout 120 python -m pytest tests/test_dropin_scripts.py -q -k two_ranks > /dev/null 2>&1 &); sleep 3; timeout 60 $P 10 2 | tee -a $O/probe.log
done
echo "== next to hog_conv_loop (level-1 conv_f16x2)" | tee -a $O/probe.log
rm -f /tmp/hog_ready; (SHAPE=64,64,64,1024,3,8 SECS=20 READY_FILE=/tmp/hog_ready timeout 60 python scripts/hog_conv_loop.py > /dev/null 2>&1 &)
for i in $(seq 1 40); do [ -f /tmp/hog_ready ] && break; sleep 0.5; done
timeout 60 $P 8 2 | tee -a $O/probe.log
