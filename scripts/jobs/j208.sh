#!/bin/bash
# round 4: full GPU suite + smoke + bench lines (default, fp16 mode, A/B against 64-channel tiles everywhere) on the wide-tile code (Cin <= 128)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j208; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2700 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2; do
  for m in 64 auto; do
    if [ $m = 64 ]; then export R2DM_F2_CO_TILE=64; else unset R2DM_F2_CO_TILE; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tile=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
    timeout 300 python $R/bench.py $A --precision fp16 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 tile=$m', j['ms_per_step'], j['value'], j.get('roofline',{}).get('frac'))"
  done
done | tee $O/ab.log
