#!/usr/bin/env python3
"""GPU-box validation of the north-star statement: the full 256-step DDPM sampler at 64x1024 against the oracle (the
reference's arithmetic, torch fp32 on the host CPU) on an identical noise tape; per-pixel delta of the FINAL sample."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from oracle import r2dm_oracle as O

S = int(os.environ.get("STEPS", "256"))
MODE = os.environ.get("MODE", "ddpm")                                    # ddpm | ddim (eta 0)
H, W = (int(v) for v in os.environ.get("RES", "64x1024").split("x"))    # 64x1024 | 128x2048 (BASELINE configs[4] geometry)
dev = "cuda"
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(H, W))
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=1, precision=os.environ.get("PRECISION", "fp32"))
rng = r2dm_amd.setup_rng([11], dev)
tape = [ddpm.randn(1, 2, H, W, rng=rng, device=dev) for _ in range(S + 1)]
it = iter(tape)
ddpm.randn = lambda *shape, rng=None, **kw: next(it)
t0 = time.time(); got = ddpm.sample(batch_size=1, num_steps=S, progress=False, rng=None, mode=MODE).cpu(); t_hip = time.time() - t0
sd = O.strip_prefix(ck["ema_weights"]); cfg = O.UNetConfig(resolution=(H, W))
torch.set_num_threads(min(32, os.cpu_count() or 1))
t0 = time.time()
want = O.sample_continuous(lambda x, c: O.unet_forward(sd, cfg, x, c), (1, 2, H, W), S, noises=[z.cpu() for z in tape], mode=MODE)
t_cpu = time.time() - t0
d = (got - want).abs().flatten().double()
print(f"{S}-step {MODE.upper()}, {H}x{W}, batch 1: HIP {t_hip:.1f} s, CPU oracle {t_cpu:.1f} s; final sample |hip - oracle|: "
      f"max {d.max():.3e}  q99.9 {torch.quantile(d, 0.999):.3e}  rms {d.pow(2).mean().sqrt():.3e}; "
      f"pixels > 1e-4: {(d > 1e-4).sum().item()} of {d.numel()}; sample range [{want.min():.3f}, {want.max():.3f}]")
