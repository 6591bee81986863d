#!/bin/bash
# round 3: conv_f16x2, the block's last tile split between multipliers and stagers: identical output? tests, timeline, A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j106; mkdir -p $O
cd $R
for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so; do for prec in fp32 fp16; do R2DM_HIP_LIB=$R/$lib PREC=$prec python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, hashlib, r2dm_amd
from r2dm_amd import synthetic
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8, precision=os.environ["PREC"])
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
y = ddpm.model(x, c); s = ddpm.sample(batch_size=3, num_steps=3, progress=False, rng=r2dm_amd.setup_rng([0, 1, 2], "cuda"))
print(os.environ["R2DM_HIP_LIB"].split("/")[-1], os.environ["PREC"], hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(s.cpu().numpy().tobytes()).hexdigest()[:16])
PY
done; done | tee $O/hash.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_range.py tests/test_hip_fp16_mode.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.log
for sh in U3_128_128 L1_64_64; do
B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=2000 SHAPES=$sh timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$sh.log; head -3 $O/tl_$sh.log; tail -7 $O/tl_$sh.log
done
cd /tmp
for rep in 1 2 3; do for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so; do
R2DM_HIP_LIB=$R/$lib timeout 300 python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$lib:', round(j['ms_per_step'],3), r['board']['sclk_mhz'], r['board']['board_w'], round(r['dominant_kernel']['ms_per_step'],3))"; done; done 2>&1 | tee $O/ab.log
