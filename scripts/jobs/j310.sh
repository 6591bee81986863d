#!/bin/bash
# round 5: fp16 mode on the 32-channel tile + the whole fp16-mode test file + bench --precision fp16 at batch 8 (u_block4 on 32-channel tiles, one plane)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j310; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_fp16_mode.py tests/test_hip_kernels.py -q -x -k "fp16 or one_product or conv3x3_both or mode" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for m in 0 1; do
R2DM_F2_NARROW=$m timeout 600 python bench.py --precision fp16 --no-cpu-baseline --no-torch-baseline --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 narrow=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
done | tee $O/fp16.log
