#!/usr/bin/env python3
"""GPU-box diagnostic (not a test): error of the HIP path and of the fp32 oracles against the fp64
oracle, for the U-Net and for teacher-forced sampling.  Prints a small table."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd  # noqa: E402
from oracle import r2dm_oracle as O  # noqa: E402
from r2dm_amd import synthetic  # noqa: E402

DEV = "cuda"
res = tuple(int(v) for v in os.environ.get("RES", "64,1024").split(","))
S = int(os.environ.get("STEPS", "8"))
mode = os.environ.get("MODE", "ddpm")
B = 1

ck = synthetic.synthetic_checkpoint(seed=0, resolution=res)
ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=B)
sd32 = O.strip_prefix(ck["ema_weights"])
sd64g = {k: v.double().to(DEV) for k, v in sd32.items()}
sd32g = {k: v.to(DEV) for k, v in sd32.items()}
cfg = O.UNetConfig(resolution=res)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 2, *res, generator=g)

print(f"== U-Net {res} ==")
for c in (-15.0, -7.0, 0.0, 7.0, 15.0):
    cond = torch.full((B,), c)
    hip = ddpm.model(x.to(DEV), cond.to(DEV)).double().cpu()
    r64 = O.unet_forward(sd64g, cfg, x.double().to(DEV), cond.double().to(DEV)).cpu()
    r32c = O.unet_forward(sd32, cfg, x, cond).double()
    r32g = O.unet_forward(sd32g, cfg, x.to(DEV), cond.to(DEV)).double().cpu()
    e = lambda a: (a - r64).abs().max().item()
    print(f"cond {c:6.1f}: |out|max {r64.abs().max():.3f}  hip {e(hip):.2e}  cpu-fp32 {e(r32c):.2e}  gpu-fp32(MIOpen) {e(r32g):.2e}")

print(f"== sampling {res}, {S} steps {mode}, same noise tape; max|x - x_fp64| per step ==")
noises = [torch.randn(B, 2, *res, generator=g) for _ in range(S + 1)]
ref = O.sample_continuous(lambda a, c: O.unet_forward(sd64g, cfg, a, c), (B, 2, *res), S, noises=noises,
                          return_all=True, mode=mode, device=DEV, dtype=torch.float64).cpu()
c32 = O.sample_continuous(lambda a, c: O.unet_forward(sd32, cfg, a, c), (B, 2, *res), S, noises=noises,
                          return_all=True, mode=mode).double()
tape = list(noises)
ddpm.randn = lambda *shape, rng=None, **kw: tape.pop(0).to(DEV)
hip = ddpm.sample(B, S, progress=False, return_all=True, mode=mode).double().cpu()
for i in range(S + 1):
    print(f"step {i}: hip {(hip[i]-ref[i]).abs().max():.2e}  cpu-fp32 {(c32[i]-ref[i]).abs().max():.2e}  hip-vs-cpu32 {(hip[i]-c32[i]).abs().max():.2e}")
