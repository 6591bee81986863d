#!/usr/bin/env python3
"""GPU-box probe: time N denoiser forwards at batch 8 with the range check deferred (and its verdict ignored) -- for timing ablations
whose libraries compute garbage (scripts/jobs/j129.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from r2dm_amd.diffusion import _range_guard
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device="cuda", show_info=False, max_batch=8)
g = torch.Generator(device="cuda").manual_seed(1); x = torch.randn(8, 2, 64, 1024, device="cuda", generator=g); c = torch.linspace(-5, 5, 8, device="cuda")
N = int(os.environ.get("N", "200"))
try:
    with _range_guard(ddpm.model):
        for _ in range(60): ddpm.model(x, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): ddpm.model(x, c)
        e1.record(); torch.cuda.synchronize()
        print(os.environ.get("R2DM_HIP_LIB", "default").split("/")[-1], "ms per forward:", round(e0.elapsed_time(e1) / N, 4))
except Exception as e:
    print("(range check:", str(e)[:60], ")")
