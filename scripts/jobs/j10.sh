#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j10; mkdir -p $O
cd $R
R2DM_DUO_MIN=100000000 timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu > $O/test.log 2>&1; tail -3 $O/test.log
echo "== pair/stream (no duo)" >> $O/abl.log
R2DM_DUO_MIN=100000000 SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128,L3_256_256,L4_512_512,L4_256_256 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
echo "== duo forced" >> $O/abl.log
R2DM_DUO_MIN=1 SHAPES=L1_64_64,L1_128_64,L1_64_128,L2_128_128 ITERS=20 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $O/abl.log
cat $O/abl.log
R2DM_DUO_MIN=100000000 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline > $O/bench_pair.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline > $O/bench_default.json 2>/dev/null
python -c "
import json
for f in ('bench_pair','bench_default'):
    j=json.load(open('$O/'+f+'.json')); print(f, j['value'], j['ms_per_step'], j['roofline']['frac'])
"
