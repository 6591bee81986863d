#!/bin/bash
# round 3: the forward's conditioning (time embedding + AdaGN projections) on an internal side stream: same output? tests; A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j118; mkdir -p $O
cd $R
for m in 0 1; do R2DM_EMBED_STREAM=$m python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8)
h = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:16]
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g)
ys = [h(ddpm.model(x, torch.linspace(-5 + i, 5, 8, device="cuda"))) for i in range(4)]   # back-to-back forwards with different conditioning
a = ddpm.sample(batch_size=3, num_steps=12, progress=False, rng=r2dm_amd.setup_rng([0, 1, 2], "cuda"))
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    s2.wait_stream(torch.cuda.default_stream()); b = h(ddpm.model(x, torch.linspace(-5, 5, 8, device="cuda")))   # a non-default caller stream
print("R2DM_EMBED_STREAM", os.environ["R2DM_EMBED_STREAM"], ys, h(a), b)
PY
done | tee $O/hash.log
timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_configs.py tests/test_hip_range.py tests/test_hip_fp16_mode.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
cd /tmp
for rep in 1 2 3; do for m in 0 1; do
R2DM_EMBED_STREAM=$m timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('conditioning on a side stream $m:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'])"; done; done 2>&1 | tee $O/ab.log
