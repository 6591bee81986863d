#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j12; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu > $O/test_kernels.log 2>&1; tail -3 $O/test_kernels.log
R2DM_DUO_MIN=1 timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv or group_norm" > $O/test_duo.log 2>&1; tail -2 $O/test_duo.log
SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128,L3_256_256,L4_512_512,L4_256_256 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/conv.log; cat $O/conv.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline > $O/bench.json 2>/dev/null
python -c "
import json
j=json.load(open('$O/bench.json')); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'])
"
timeout 1200 python -m pytest tests/test_hip_unet.py -x -q -m gpu > $O/test_unet.log 2>&1; tail -5 $O/test_unet.log
