#!/bin/bash
# round 5, last session: the driver's N = 2 launch line on a ONE-GPU box (both ranks land on cuda:0) with the RCCL backend -- does an N > 1 RCCL group come up at all when two ranks share a device,
# and if it refuses: does it fail fast and loudly (not hang)?  Own short timeout.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j380; mkdir -p $O; cd $R
NCCL_DEBUG=WARN timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 > $O/n2.json 2> $O/n2.err
echo "rc $?" | tee $O/rc.log
grep -i "duplicate\|NCCL WARN\|ncclInvalid\|Error" $O/n2.err | head -8 | cut -c1-300 | tee -a $O/rc.log
tail -1 $O/n2.json | cut -c1-600 | tee -a $O/rc.log
