#!/bin/bash
# round 5: v_cndmask with an SGPR-pair mask in a hot loop (probe mode 4), alone and next to the sampler process (the neighbour the branch-free fir_up2 fails beside)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j367; mkdir -p $O; cd $R
P=$R/build_probe/dpp_shift_probe
echo "== alone" | tee $O/probe.log; timeout 60 $P 5 4 | tee -a $O/probe.log
echo "== next to the sampler loop" | tee -a $O/probe.log
rm -f /tmp/hs_ready; (SECS=30 READY_FILE=/tmp/hs_ready timeout 90 python scripts/hog_sampler_loop.py > /dev/null 2>&1 &)
for i in $(seq 1 60); do [ -f /tmp/hs_ready ] && break; sleep 0.5; done; sleep 2
timeout 60 $P 10 4 | tee -a $O/probe.log
timeout 60 $P 8 2 | tee -a $O/probe.log
