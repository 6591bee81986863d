#!/bin/bash
# round 3: full GPU suite + default bench line on the current library (tile-order alternation, other_configs)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j92; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
cd /tmp; python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; j=json.load(open('$O/bench_n1.json')); r=j['roofline']; print(round(j['value'],3), round(j['ms_per_step'],3), 'vs', round(j['vs_baseline'],2), 'exact', round(j['exact_split_baseline']['value'],3), 'frac', round(r['frac'],3), {k: round(v['value'],2) for k,v in j['other_configs'].items()}, 'cpu', j['cpu_baseline']['value'])"
