#!/bin/bash
# round 3: per-step time inside one long sample() call; the prize of folding gn_finalize away (upper bound, wrong results)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j83; mkdir -p $O
cd $R
{
STEPS=256 timeout 300 python scripts/step_times.py
STEPS=64 timeout 300 python scripts/step_times.py
} 2>&1 | grep -v amdgpu.ids | tee $O/step_times.log
b() { timeout 300 env $1 python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline 2>$O/err.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1:', round(j['ms_per_step'],3), 'ms/step sclk', r['board']['sclk_mhz'], r['board']['board_w'])" || tail -3 $O/err.log; }
for rep in 1 2 3; do b X=1; b R2DM_SKIP_FINALIZE=1; done 2>&1 | tee $O/finalize_prize.log
