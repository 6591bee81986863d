#!/bin/bash
# final check of the round's last code: build() as the driver runs it, smoke, the whole GPU suite, the default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j347; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | tail -3
cd /tmp; timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['vs_baseline'])"
