#!/usr/bin/env python3
"""GPU-box probe: the fixed cost of one ddpm.sample() call (bench.py times ONE call of --steps reverse steps and scales it): wall time of calls of 1 .. 64 steps, a line through
them, and where the host waits inside a 16-step call (torch profiler: cudaStreamSynchronize / memcpy)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
dev = torch.device("cuda", 0)
B = 8
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device=dev, show_info=False, max_batch=B)
run = lambda n: ddpm.sample(batch_size=B, num_steps=n, progress=False, rng=r2dm_amd.setup_rng(list(range(B)), dev))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    run(8); torch.cuda.synchronize()
pts = []
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    run(n); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); run(n); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    pts.append((n, best * 1e3)); print(f"sample({n:3d} steps): {best * 1e3:8.3f} ms  = {best * 1e3 / n:6.3f} ms/step", flush=True)
import numpy as np
x, y = np.array([p[0] for p in pts[3:]], float), np.array([p[1] for p in pts[3:]])
a, b = np.polyfit(x, y, 1)
print(f"fit over 8..128 steps: {a:.3f} ms/step + {b:.3f} ms per call")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(16); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
