#!/bin/bash
# round 4: what the board does with the denser MFMA stream: 256-step timed calls (shader clock and board power sampled over the timed region)
# with 64-channel tiles everywhere / the round-4 rule (Cin <= 128) / 128-channel tiles wherever the shape allows, two alternations
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j216; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 256 --warmup 8"
for i in 1 2; do
  for m in 64 auto 128; do
    if [ $m = auto ]; then unset R2DM_F2_CO_TILE; else export R2DM_F2_CO_TILE=$m; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j['roofline'].get('board') or {}; d=j['roofline']['dominant_kernel']
print('co_tile=$m  %.3f ms/step  %.3f images/s  conv_f16x2 avg launch %.1f us  frac %.3f  sclk %s MHz  board %s W' % (j['ms_per_step'], j['value'], d['avg_launch_us'], j['roofline']['frac'], b.get('sclk_mhz'), b.get('board_w')))"
  done
done | tee $O/clock_ab.log
