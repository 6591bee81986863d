#!/bin/bash
# round 5: is the scalar data cache isolated between processes?  (scripts/probes/dpp_shift_probe.hip mode 3) two instances with different tags, tables at the same virtual address
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j365; mkdir -p $O; cd $R
P=$R/build_probe/dpp_shift_probe
echo "== alone" | tee $O/probe.log; timeout 60 $P 5 3 0x11111111 | tee -a $O/probe.log
echo "== two instances, different tags" | tee -a $O/probe.log; (timeout 60 $P 10 3 0x22222222 > $O/second.log &) ; sleep 1; timeout 60 $P 6 3 0x11111111 | tee -a $O/probe.log; sleep 5; cat $O/second.log | tee -a $O/probe.log
echo "== three instances" | tee -a $O/probe.log; (timeout 60 $P 10 3 0x33333333 > $O/third.log &); (timeout 60 $P 10 3 0x22222222 > $O/second.log &) ; sleep 1; timeout 60 $P 6 3 0x11111111 | tee -a $O/probe.log; sleep 5; cat $O/second.log $O/third.log | tee -a $O/probe.log
