#!/bin/bash
# round 3: 1x1 convolutions with Cout % 256 == 0 through 256-channel blocks (proj_tall_f16x2_kernel): identical output? tests; kernel times; A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j124; mkdir -p $O
cd $R
for m in 0 1; do for prec in fp32 fp16; do R2DM_PROJ_TALL=$m PREC=$prec python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8, precision=os.environ["PREC"])
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
y = ddpm.model(x, c)
print("R2DM_PROJ_TALL", os.environ["R2DM_PROJ_TALL"], os.environ["PREC"], hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
PY
done; done | tee $O/hash.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_fp16_mode.py tests/test_hip_range.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
cd /tmp
for m in 0 1; do
rm -rf /tmp/prof_$m
R2DM_PROJ_TALL=$m timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -o p -- python $R/bench.py --steps 4 --warmup 1 --prewarm-s 0.3 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$m.json 2>/dev/null
f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
echo "R2DM_PROJ_TALL=$m: $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'proj_' in r['Name'] and 'pack' not in r['Name']: print(r['Name'][11:46], 'calls', r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us;', end=' ')
")"
done 2>&1 | tee $O/kernels.log
for rep in 1 2 3; do for m in 0 1; do
R2DM_PROJ_TALL=$m timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('tall projection blocks $m:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'])"; done; done 2>&1 | tee $O/ab.log
