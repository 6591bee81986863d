#!/bin/bash
# round 4: final check of the committed tree: smoke + GPU suite + default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j222; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.log
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp; timeout 600 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; j=json.load(open('$O/bench_n1.json')); print(round(j['value'],3), round(j['ms_per_step'],3), j['vs_baseline'], j['roofline']['frac'], j['cpu_baseline']['value'])"
