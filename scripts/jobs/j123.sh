#!/bin/bash
# round 3: attention loop reordered (scores of the next tile issued behind the barrier): identical output? tests; kernel times; A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j123; mkdir -p $O
cd $R
for lib in build_probe/lib_head.so r2dm_amd/libr2dm_hip.so; do for prec in fp32 fp16; do R2DM_HIP_LIB=$R/$lib PREC=$prec python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8, precision=os.environ["PREC"])
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
y = ddpm.model(x, c)
d2, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0, resolution=(16, 128)), device="cuda", show_info=False, max_batch=3, precision=os.environ["PREC"])
y2 = d2.model(torch.randn(3, 2, 16, 128, device="cuda", generator=g), torch.linspace(-5, 5, 3, device="cuda"))   # one key tile only
print(os.environ["R2DM_HIP_LIB"].split("/")[-1], os.environ["PREC"], hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(y2.cpu().numpy().tobytes()).hexdigest()[:16])
PY
done; done | tee $O/hash.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_fp16_mode.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
cd /tmp
for lib in build_probe/lib_head.so r2dm_amd/libr2dm_hip.so; do
n=$(basename $lib .so); rm -rf /tmp/prof_$n
R2DM_HIP_LIB=$R/$lib timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $R/bench.py --steps 4 --warmup 1 --prewarm-s 0.3 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$n.json 2>/dev/null
f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
echo "$n: $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'attention' in r['Name']: print(r['Name'][11:40], round(float(r['AverageNs'])/1e3,1), 'us;', end=' ')
")"
done 2>&1 | tee $O/kernels.log
for rep in 1 2 3; do for lib in build_probe/lib_head.so r2dm_amd/libr2dm_hip.so; do
R2DM_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'])"; done; done 2>&1 | tee $O/ab.log


