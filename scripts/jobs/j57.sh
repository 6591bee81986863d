#!/bin/bash
# noise drawn on a side stream ahead of the denoiser + time embedding / AdaGN projections on the handle's side stream: tests, A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j57; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
R2DM_NOISE_STREAM=$1 R2DM_TEMB_STREAM=$2 timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench noise_stream=$1 temb_stream=$2', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"; done; done
