#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j94; mkdir -p $O
cd $R; timeout 900 python scripts/soak.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.log
cd /tmp; python $R/bench.py --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.load(open('$O/bench.json')); t=j['torch_rocm_baseline']; print(round(j['value'],3), 'torch fp32', round(t['value'],3), 'autocast', t.get('fp16_autocast'))"
python $R/bench.py --precision fp16 --no-cpu-baseline --no-other-configs > $O/bench_fp16.json 2> $O/bench_fp16.err; python -c "
import json; j=json.load(open('$O/bench_fp16.json')); t=j['torch_rocm_baseline']; print('fp16 mode', round(j['value'],3), 'vs', j['vs_baseline'], 'autocast', t.get('fp16_autocast'))"
