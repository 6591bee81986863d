#!/bin/bash
# round 3: why is a 256-step sample() call 5 % slower per step than a 20-step call of the same job?  Host run-ahead bound.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j82; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
b() { timeout 300 env R2DM_STEPS_IN_FLIGHT=$1 python bench.py --steps $2 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline 2>$O/err.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('in_flight $1 steps $2:', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'img/s sclk', r['board']['sclk_mhz'], r['board']['board_w'])" || tail -3 $O/err.log; }
for rep in 1 2; do
b 0 20; b 0 64; b 0 256
b 4 20; b 4 256
b 2 256; b 8 256; b 16 256
done 2>&1 | tee $O/ahead.log
