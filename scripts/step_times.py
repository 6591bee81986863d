#!/usr/bin/env python3
"""GPU-box probe: per-step GPU time inside ONE sample() call (events after every posterior update), printed as averages over
groups of 16 steps, with the board clock / power sampled alongside.  Why is a 256-step call slower per step than a 64-step one?"""
import os, sys, time, threading, glob
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from bench import BoardSampler
dev = torch.device("cuda", 0)
B = 8
S = int(os.environ.get("STEPS", "256"))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=B)
def run(steps): return ddpm.sample(batch_size=B, num_steps=steps, progress=False, rng=r2dm_amd.setup_rng(list(range(B)), dev))
t0 = time.time()
while time.time() - t0 < 2.5: run(8); torch.cuda.synchronize()
post = ddpm._posterior
for rep in range(2):
    evs = []
    def hook(*a, **k):
        r = post(*a, **k); e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); return r
    ddpm._posterior = hook
    e0 = torch.cuda.Event(enable_timing=True); torch.cuda.synchronize()
    with BoardSampler(0) as bs:
        e0.record(); run(S); torch.cuda.synchronize()
    ddpm._posterior = post
    ts = [e0.elapsed_time(e) for e in evs]
    per = [ts[0]] + [b - a for a, b in zip(ts, ts[1:])]
    g = 16
    print(f"rep {rep}: {S} steps, total {ts[-1]:.1f} ms, mean {ts[-1]/S:.3f} ms/step; per-group-of-{g} means:", " ".join(f"{sum(per[i:i+g])/len(per[i:i+g]):.2f}" for i in range(0, S, g)), flush=True)
    clk = [a / 1e6 for a, _ in bs.samples if a]; pw = [b / 1e6 for _, b in bs.samples if b]
    k = max(1, len(clk) // 8)
    print("   sclk MHz over the call:", " ".join(f"{sum(clk[i:i+k])/len(clk[i:i+k]):.0f}" for i in range(0, len(clk), k)), "| W:", " ".join(f"{sum(pw[i:i+k])/len(pw[i:i+k]):.0f}" for i in range(0, len(pw), k)), flush=True)
