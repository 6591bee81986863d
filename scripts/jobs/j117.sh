#!/bin/bash
# round 3: the step's noise drawn on a side stream before the denoiser call: same samples? sampler tests; A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j117; mkdir -p $O
cd $R
for m in 0 1; do R2DM_NOISE_STREAM=$m python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8)
h = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:16]
a = ddpm.sample(batch_size=3, num_steps=12, progress=False, rng=r2dm_amd.setup_rng([0, 1, 2], "cuda"))
b = ddpm.sample(batch_size=2, num_steps=5, progress=False, rng=torch.Generator(device="cuda").manual_seed(3), mode="ddim", ddim_eta=0.5)
torch.manual_seed(9); c = ddpm.sample(batch_size=2, num_steps=4, progress=False, return_all=True)
print("R2DM_NOISE_STREAM", os.environ["R2DM_NOISE_STREAM"], h(a), h(b), h(c))
PY
done | tee $O/hash.log
timeout 1200 python -m pytest tests/test_hip_unet.py tests/test_hip_configs.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
cd /tmp
for rep in 1 2 3; do for m in 0 1; do
R2DM_NOISE_STREAM=$m timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('noise on a side stream $m:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'])"; done; done 2>&1 | tee $O/ab.log
