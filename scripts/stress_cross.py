#!/usr/bin/env python3
"""GPU-box probe: U-Net forwards of the tree in the CURRENT DIRECTORY (its r2dm_amd, its libr2dm_hip.so) next to an arbitrary
neighbour process (HOG_CMD, a shell command line started after the reference forward; HOG_WAIT seconds to come up).
Counts forwards that differ bitwise from the first and prints where the final output differs.  Used to tell aggressor from
victim in the shared-GPU failure of round 2 (scripts/jobs/j77.sh): the main role and the neighbour can come from different
commits (build_probe/bis_<commit>, scripts/make_tree.sh) or the neighbour can be a single kernel in a loop."""
import os, shlex, signal, subprocess, sys, time
import torch
sys.path.insert(0, os.getcwd())
import r2dm_amd
from r2dm_amd import synthetic
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "2"))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=B, precision=os.environ.get("PRECISION", "fp32"))
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, 2, 64, 1024, device=dev, generator=g); c = torch.full((B,), -3.0, device=dev)
iters = int(os.environ.get("ITERS", "150"))
ref = ddpm.model(x, c).clone()
for _ in range(20): assert torch.equal(ddpm.model(x, c), ref)  # alone: deterministic
p = None
cmd = os.environ.get("HOG_CMD", "")
if cmd:
    p = subprocess.Popen(cmd, shell=True, preexec_fn=os.setsid, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(float(os.environ.get("HOG_WAIT", "40")))
bad = 0; worst = 0.0; where = []
torch.cuda.synchronize(); t0 = time.time()
with ddpm.model.deferred_range_check():
    for i in range(iters):
        y = ddpm.model(x, c)
        if not torch.equal(y, ref):
            bad += 1; worst = max(worst, (y - ref).abs().max().item())
            if len(where) < 3:
                d = (y != ref); where.append(f"{int(d.sum())} px, rows {d.any(3).any(1).any(0).nonzero().flatten().tolist()[:12]}")
ms = (time.time() - t0) / iters * 1e3
alive = p is not None and p.poll() is None
print(f"[{os.environ.get('TAG','')}] main={os.path.basename(os.getcwd())} precision={os.environ.get('PRECISION','fp32')} algo={os.environ.get('R2DM_CONV_ALGO','default')} "
      f"neighbour={cmd[:90]!r}: {bad:3d} of {iters} forwards differ (max |diff| {worst:.2e}); {ms:.2f} ms per forward"
      f"{'' if (alive or not cmd) else '  [NEIGHBOUR NOT RUNNING AT THE END]'} {where}", flush=True)
if p is not None:
    try: os.killpg(os.getpgid(p.pid), signal.SIGTERM)
    except ProcessLookupError: pass
    p.wait()
