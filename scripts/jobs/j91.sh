#!/bin/bash
# round 3: alternating tile-walk direction of consecutive conv_f16x2 launches (Infinity Cache reuse at level 1): identical results? faster?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j91; mkdir -p $O
cd $R
for m in 0 1; do R2DM_TILE_ORDER=$m python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8)
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
y = ddpm.model(x, c); print("R2DM_TILE_ORDER", os.environ["R2DM_TILE_ORDER"], hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16], float(y.abs().mean()))
PY
done
cd /tmp
for rep in 1 2 3; do for m in 0 1; do
R2DM_TILE_ORDER=$m timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('tile order $m:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; done; done 2>&1 | tee $O/ab.log
