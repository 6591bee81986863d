#!/bin/bash
# is the whole step sensitive to the stagers' work at all?  (F2_NO_XF: wrong numbers, timing only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j49; mkdir -p $O
cd $R
{
for rep in 1 2; do
for lib in "$R/r2dm_amd/libr2dm_hip.so" "$R/build_probe/lib_f2_noxf_np.so"; do
  R2DM_HIP_LIB=$lib timeout 200 python - <<PY
import os, sys, time, torch
sys.path.insert(0, "$R")
import r2dm_amd
from r2dm_amd import synthetic
ck = synthetic.synthetic_checkpoint(seed=0)
ddpm, _, _ = r2dm_amd.setup_model(ck, device="cuda", show_info=False, max_batch=8)
net = ddpm.model
x = torch.randn(8, 2, 64, 1024, device="cuda"); c = torch.full((8,), -3.0, device="cuda")
def go(n):
    for _ in range(n): net(x, c)
    torch.cuda.synchronize()
try:
    with net.deferred_range_check():
        go(150); t0 = time.perf_counter(); go(100); dt = (time.perf_counter() - t0) / 100 * 1e3
except Exception as e:
    dt = float("nan"); print("err", e)
print("lib=%s forward %.3f ms" % (os.path.basename("$lib"), dt))
PY
done
done
} 2>&1 | grep -v amdgpu.ids | grep "lib=" | tee $O/noxf.log
