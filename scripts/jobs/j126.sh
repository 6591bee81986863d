#!/bin/bash
# round 3: fir_down starts on the samples its producer wrote last (Infinity Cache): identical output; kernel time; A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j126; mkdir -p $O
cd $R
for m in 0 1; do R2DM_FIR_ORDER=$m python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8)
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
print("R2DM_FIR_ORDER", os.environ["R2DM_FIR_ORDER"], hashlib.sha1(ddpm.model(x, c).cpu().numpy().tobytes()).hexdigest()[:16])
PY
done | tee $O/hash.log
cd /tmp
for m in 0 1; do
rm -rf /tmp/prof_$m
R2DM_FIR_ORDER=$m timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -o p -- python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0.3 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$m.json 2>/dev/null
f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
echo "R2DM_FIR_ORDER=$m: $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'fir_down' in r['Name'] or 'gn_partial' in r['Name']: print(r['Name'][6:30], round(float(r['AverageNs'])/1e3,1), 'us;', end=' ')
")"
done 2>&1 | tee $O/kernels.log
for rep in 1 2 3; do for m in 0 1; do
R2DM_FIR_ORDER=$m timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('fir_down last-first $m:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'])"; done; done 2>&1 | tee $O/ab.log
