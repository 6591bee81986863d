#!/bin/bash
# full GPU suite + smoke + bench with baselines (as the driver runs them)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j20; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tee $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json
j=json.load(open('$O/bench.json')); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('torch_rocm_baseline'), j.get('cpu_baseline'))
"
