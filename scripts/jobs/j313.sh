#!/bin/bash
# round 5: fp16 storage of the tensor between a residual block's convolutions in the one-plane mode -- tests, bench A/B against R2DM_FP16_STORAGE=0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j313; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_fp16_mode.py -q -x -s > $O/pytest.log 2>&1; tail -4 $O/pytest.log; grep "rel rms\|48-step" $O/pytest.log
for i in 1 2; do for m in 0 1; do
R2DM_FP16_STORAGE=$m timeout 600 python bench.py --precision fp16 --no-cpu-baseline --no-torch-baseline --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 storage=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
done; done | tee $O/fp16.log
