#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j47; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2; do timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"; done
