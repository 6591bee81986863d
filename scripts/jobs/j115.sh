#!/bin/bash
# round 3: GroupNorm statistics of the downsampled tensors fused into the FIR pass (3 gn_partial launches and one pass over each tensor less per step)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j115; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_range.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
for m in 0 1; do for prec in fp32; do R2DM_FIR_STATS=$m PREC=$prec python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8, precision=os.environ["PREC"])
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
y = ddpm.model(x, c); s = ddpm.sample(batch_size=3, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([0, 1, 2], "cuda"))
print("R2DM_FIR_STATS", os.environ["R2DM_FIR_STATS"], os.environ["PREC"], hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(s.cpu().numpy().tobytes()).hexdigest()[:16], float(y.double().abs().mean()))
PY
done; done | tee $O/hash.log
cd /tmp
for rep in 1 2 3; do for m in 0 1; do
R2DM_FIR_STATS=$m timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('fused FIR statistics $m:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'])"; done; done 2>&1 | tee $O/ab.log
