#!/usr/bin/env python3
"""GPU-box probe: is the U-Net forward independent of a second process running the same forward on the same GPU?
ROLE=hog runs forwards for SECS seconds; the main role repeats one forward ITERS times and counts bitwise mismatches."""
import os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "2"))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=B, precision=os.environ.get("PRECISION", "fp32"))
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, 2, 64, 1024, device=dev, generator=g); c = torch.full((B,), -3.0, device=dev)
if os.environ.get("ROLE") == "hog":
    t0 = time.time()
    with ddpm.model.deferred_range_check():
        while time.time() - t0 < float(os.environ.get("SECS", "40")):
            for _ in range(10): ddpm.model(x, c)
            torch.cuda.synchronize()
    sys.exit(0)
iters = int(os.environ.get("ITERS", "150"))
for hog in ("none", "forward"):
    p = None
    if hog != "none":
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, ROLE="hog"))
        time.sleep(25)
    ref = ddpm.model(x, c).clone()
    bad = 0; worst = 0.0
    with ddpm.model.deferred_range_check():
        for i in range(iters):
            y = ddpm.model(x, c)
            if not torch.equal(y, ref):
                bad += 1; worst = max(worst, (y - ref).abs().max().item())
    print(f"precision={os.environ.get('PRECISION','fp32')} algo={os.environ.get('R2DM_CONV_ALGO','default')} neighbour={hog:8s}: {bad:3d} of {iters} forwards differ from the first (max |diff| {worst:.2e})", flush=True)
    if p is not None:
        p.terminate(); p.wait()
