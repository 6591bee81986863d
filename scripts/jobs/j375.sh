#!/bin/bash
# the round's last code change (no packed-fp32 instruction selection in attention / in_conv / out_conv / FIR / posterior): whole GPU suite, smoke, the neighbour soaks, the default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j375; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | tail -3
MODE=process HOG_SHAPE=128,64,64,1024,1,8 REPS=1000 timeout 300 python scripts/coresidency_probe.py 2>&1 | grep coresidency_probe | tee $O/soak.log
REPS=6 timeout 600 python scripts/two_rank_diff.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/soak.log
NEIGHBOUR=conv HOG_SHAPE=128,64,64,1024,1,8 SECS=8 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep "^fir_up_soak" | tee -a $O/soak.log
cd /tmp; timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['vs_baseline'])"
