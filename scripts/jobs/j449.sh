#!/bin/bash
# final tree: smoke, the default bench line, the whole GPU suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j449; mkdir -p $O
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5 | tee $O/smoke.log
( time timeout 1200 python bench.py --gpus 1 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_time.log
python -c "
import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=j['roofline']
print('value', j['value'], 'ms', j['ms_per_step'], 'frac', r['frac'], 'band', r['event_bracket']['frac_band'], 'cpu', j['cpu_baseline']['value'], j['cpu_baseline']['unit'])" | tee $O/bench_line.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
