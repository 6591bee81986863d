#!/usr/bin/env python3
"""GPU-box probe, neighbour role: sample(2, 2 steps) at 64x1024 in a loop for SECS seconds (the neighbour next to which the branch-free fir_up2 is wrong)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, r2dm_amd
from conftest import synthetic_ckpt
m, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=(64, 1024)), device="cuda", show_info=False, max_batch=2)
if os.environ.get("READY_FILE"): open(os.environ["READY_FILE"], "w").close()
t0 = time.time()
while time.time() - t0 < float(os.environ.get("SECS", "40")):
    m.sample(2, 2, progress=False); torch.cuda.synchronize()
