#!/bin/bash
# round 5, last session: (1) where the sampler fuzz's one loose-bar miss (DDIM eta = 1, j378 case 15) comes from: c_2 per step on host / device / float64 and the product against the
# oracle with the sampler scalars evaluated on the host | on the device (scripts/ddim_eta1_scalars.py); (2) the default bench line after the vs_baseline -> vs_torch_rocm_baseline rename
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j379; mkdir -p $O; cd $R
timeout 300 python scripts/ddim_eta1_scalars.py 2>&1 | grep "ddim_eta1\|Error\|Traceback" -A3 | tee $O/ddim_eta1.log
cd /tmp; timeout 600 python $R/bench.py --no-cpu-baseline --no-compile-baseline --no-other-configs > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['vs_baseline'], j['vs_torch_rocm_baseline'])" | tee $O/bench.log
