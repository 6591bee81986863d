#!/bin/bash
# operand pre-pass with folded GroupNorm, level 2 of the switch: also d_block4's 512 -> 512 launches (64-channel tiles, eight per x tile); bit-identity, then A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j430; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_unet.py -m gpu -x -q -k operand_prepass 2>&1 | tail -5 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for m in 0 2 1; do
    R2DM_F2_PRESPLIT_NARROW=$m timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$m.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench narrow_presplit=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp && R2DM_F2_PRESPLIT_NARROW=2 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p2 -- python $R/bench.py $A --steps 16 --warmup 2 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}' | cut -c1-200 | tee $O/stats_head.txt
