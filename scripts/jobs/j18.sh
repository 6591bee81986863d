#!/bin/bash
# two INDEPENDENT processes sharing the GPU (no torch.distributed): is the forward self-consistent in each?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j18; mkdir -p $O; rm -f $O/cmp.log
cd $R
cat > /tmp/selfcheck.py <<'PY'
import os, sys, torch, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import r2dm_amd
from r2dm_amd import synthetic
tag = sys.argv[1]
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device="cuda", show_info=False, max_batch=2)
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(2, 2, 64, 1024, device="cuda", generator=g); c = torch.tensor([-15.0, -15.0], device="cuda")
ref = ddpm.model(x, c).clone()
bad = 0; worst = 0.0
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[2]):
    y = ddpm.model(x, c)
    d = (y - ref).abs().max().item(); n += 1
    if d > 0: bad += 1; worst = max(worst, d)
print(f"{tag}: {n} forwards, {bad} differ from the first one, worst |diff| {worst:.3e}", flush=True)
PY
echo "--- one process alone" | tee -a $O/cmp.log
python /tmp/selfcheck.py solo 6 2>&1 | grep -v amdgpu | tee -a $O/cmp.log
echo "--- two processes at once" | tee -a $O/cmp.log
python /tmp/selfcheck.py A 10 > /tmp/A.log 2>&1 &
python /tmp/selfcheck.py B 10 > /tmp/B.log 2>&1 &
wait
grep -h forwards /tmp/A.log /tmp/B.log | tee -a $O/cmp.log
echo "--- two processes at once, round-1 library" | tee -a $O/cmp.log
R2DM_HIP_LIB=$R/build_probe/lib_r1.so python /tmp/selfcheck.py A 10 > /tmp/A.log 2>&1 &
R2DM_HIP_LIB=$R/build_probe/lib_r1.so python /tmp/selfcheck.py B 10 > /tmp/B.log 2>&1 &
wait
grep -h forwards /tmp/A.log /tmp/B.log | tee -a $O/cmp.log
echo "--- two processes at once, fp32-MFMA algorithm only" | tee -a $O/cmp.log
R2DM_CONV_ALGO=f32 python /tmp/selfcheck.py A 10 > /tmp/A.log 2>&1 &
R2DM_CONV_ALGO=f32 python /tmp/selfcheck.py B 10 > /tmp/B.log 2>&1 &
wait
grep -h forwards /tmp/A.log /tmp/B.log | tee -a $O/cmp.log
