#!/bin/bash
# round 6 evidence set r06d on the final code (tile-end diet + batched loads in the FIR resamplers and the attention cores): scripts/jobs/j306.sh, then the round-5 tree against this tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
JOB=j442 bash $R/scripts/jobs/j306.sh
O=$R/gpurun_out/j442
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for t in build_probe/r05tree .; do
    (cd $R/$t && timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tree=$t', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))")
  done
done | tee $O/ab_r05_r06.log
