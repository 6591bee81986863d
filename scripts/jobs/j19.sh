#!/bin/bash
# GPU sharing: my forward loop next to an UNRELATED compute process (torch matmuls)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j19; mkdir -p $O; rm -f $O/cmp.log
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
cat > /tmp/other.py <<'PY'
import torch, time, sys
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1]):
    c = a @ b; n += 1
    if n % 50 == 0: torch.cuda.synchronize()
torch.cuda.synchronize(); print("other: matmuls", n, flush=True)
PY
echo "--- forward loop next to an unrelated matmul process" | tee -a $O/cmp.log
python /tmp/other.py 14 > /tmp/B.log 2>&1 &
sleep 3
python /tmp/selfcheck.py A 8 > /tmp/A.log 2>&1
wait
grep -h "forwards\|other" /tmp/A.log /tmp/B.log | tee -a $O/cmp.log
