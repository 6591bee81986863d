#!/usr/bin/env python3
"""GPU-box timing probe: ms per U-Net forward and per reverse step for a few batch sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic

dev = "cuda"
res = tuple(int(v) for v in os.environ.get("RES", "64,1024").split(","))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=res)
for B in [int(b) for b in os.environ.get("BATCHES", "1,8").split(",")]:
    ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=B, precision=os.environ.get("PRECISION", "fp32"))
    x = torch.randn(B, 2, *res, device=dev)
    c = torch.zeros(B, device=dev)
    t0 = time.perf_counter(); y = ddpm.model(x, c); torch.cuda.synchronize()
    print(f"B={B}: first forward (incl. pack) {time.perf_counter()-t0:.2f}s", flush=True)
    n = int(os.environ.get("ITERS", "5"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = ddpm.model(x, c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"B={B}: forward {ms:.2f} ms  -> {B*234.52e9*(res[0]*res[1]/65536)/ms/1e9:.1f} TFLOP/s ({B*234.52e9*(res[0]*res[1]/65536)/ms/1e9/157.3*100:.1f}% of fp32 peak)", flush=True)
    t0 = time.perf_counter(); out = ddpm.sample(B, 4, progress=False, rng=r2dm_amd.setup_rng(list(range(B)), dev)); torch.cuda.synchronize()
    print(f"B={B}: sample 4 steps {(time.perf_counter()-t0)/4*1e3:.2f} ms/step", flush=True)
    del ddpm
