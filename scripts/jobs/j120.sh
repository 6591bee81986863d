#!/bin/bash
# round 3: ada_proj with eight samples per pass: tests (golden AdaGN projections), same samples?, kernel time
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j120; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
for lib in build_probe/lib_dc_u4.so r2dm_amd/libr2dm_hip.so; do R2DM_HIP_LIB=$R/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
h = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:16]
out = []
for mb in (8, 3, 11):
    ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0, resolution=(16, 128)), device="cuda", show_info=False, max_batch=mb)
    g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(mb, 2, 16, 128, device="cuda", generator=g)
    out.append(h(ddpm.model(x, torch.linspace(-5, 5, mb, device="cuda"))))
print(os.environ["R2DM_HIP_LIB"].split("/")[-1], out)
PY
done | tee $O/hash.log
cd /tmp
for lib in build_probe/lib_dc_u4.so r2dm_amd/libr2dm_hip.so; do
n=$(basename $lib .so); rm -rf /tmp/prof_$n
R2DM_HIP_LIB=$R/$lib timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0.5 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$n.json 2>/dev/null
f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
echo "$n: $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'ada_proj' in r['Name'] or 'time_' in r['Name']: print(r['Name'][11:32], round(float(r['AverageNs'])/1e3,1), 'us;', end=' ')
")"
done 2>&1 | tee $O/variants.log
