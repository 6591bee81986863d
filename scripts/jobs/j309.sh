#!/bin/bash
# round 5: what do the nine noise launches of a step cost the step?  default (side stream) | main stream | no draws at all (R2DM_DEBUG_FIXED_NOISE=1: wrong samples, timing only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j309; mkdir -p $O
cd $R
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in side main none; do
    case $m in side) E="";; main) E="R2DM_NOISE_STREAM=0";; none) E="R2DM_DEBUG_FIXED_NOISE=1";; esac
    env $E timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench noise=$m', round(j['ms_per_step'],3), round(j['value'],3))"
  done
done | tee $O/ab_noise.log
