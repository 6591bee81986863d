#!/usr/bin/env python3
"""GPU-box soak: (a) the same seeded 128-step sample() call three times -- bitwise equal; (b) 600 forwards at batch 8 against the first, alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
dev = torch.device("cuda", 0)
ddpm, _, _ = r2dm_amd.setup_model(synthetic.synthetic_checkpoint(seed=0), device=dev, show_info=False, max_batch=8)
for prec in ("fp32", "fp16", "fp32-bf16x3"):
    ddpm.model.set_precision(prec)
    S = 128 if prec == "fp32" else 32
    outs = [ddpm.sample(batch_size=8, num_steps=S, progress=False, rng=r2dm_amd.setup_rng(list(range(8)), dev)) for _ in range(3)]
    print(f"{prec}: three {S}-step sample() calls, batch 8: bitwise equal = {torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])}", flush=True)
    x = torch.randn(8, 2, 64, 1024, device=dev); c = torch.linspace(-6, 6, 8, device=dev)
    ref = ddpm.model(x, c).clone(); bad = 0
    with ddpm.model.deferred_range_check():
        for i in range(200 if prec != "fp32" else 600):
            bad += int(not torch.equal(ddpm.model(x, c), ref))
    print(f"{prec}: forwards differing from the first: {bad}", flush=True)
