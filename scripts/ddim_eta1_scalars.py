#!/usr/bin/env python3
"""Diagnostic behind the one loose-bar miss of the sampler fuzz (scripts/jobs/j378.sh, case 15: eps, DDIM eta = 1, cosine, 4 steps): the reference's
c_2 = sqrt(1 - alpha_s^2 - c_1^2) (/root/reference/models/diffusion/continuous_time.py:224) is rounding noise on the last step for eta = 1.  The product
evaluates the step scalars on the host (bit-identical to the reference's CPU run, tests/test_host.py); the fuzz's oracle evaluates them with the device's
libm.  This script prints c_2 of every step in float32 on the host, float32 on the device and float64, and compares the product's sample with the oracle
run BOTH ways on one noise tape (denoiser: the fp64 oracle U-Net on the device in both)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from oracle import r2dm_oracle as O
dev = torch.device("cuda", 0)
RES, S, B, ETA = (16, 128), 4, 4, 1.0
for where, dt in (("host float32", (torch.device("cpu"), torch.float32)), ("device float32", (dev, torch.float32)), ("host float64", (torch.device("cpu"), torch.float64))):
    st = torch.linspace(1.0, 0.0, S + 1, device=dt[0]).to(dt[1])
    row = []
    for i in range(S):
        a_t, s_t = O.alpha_sigma(O.log_snr_cosine(st[i:i + 1])); a_s, s_s = O.alpha_sigma(O.log_snr_cosine(st[i + 1:i + 2]))
        c1 = ETA * s_s / s_t * (1 - a_t**2 / a_s**2).sqrt(); row.append((1 - a_s**2 - c1**2).sqrt().item())
    print(f"ddim_eta1 c_2 per step, {where:15s}:", " ".join(f"{v:.6e}" for v in row), flush=True)
ck = synthetic.synthetic_checkpoint(seed=0, resolution=RES, prediction_type="eps", noise_schedule="cosine")
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=8, precision="fp32")
mk = lambda: r2dm_amd.setup_rng(list(range(100, 100 + B)), dev)
got = ddpm.sample(batch_size=B, num_steps=S, progress=False, rng=mk(), mode="ddim", ddim_eta=ETA)
sd = {k: v.double().to(dev) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
cfg = O.UNetConfig(resolution=RES)
net = lambda x, c: O.unet_forward(sd, cfg, x.to(dev).double(), c.to(dev).double()).float()
g = mk(); tape = [O.draw_noise((B, 2, *RES), g, dev, torch.float32) for _ in range(S + 1)]  # the draws of the generators the product used
for where, d in (("device", dev), ("host", torch.device("cpu"))):
    want = O.sample_continuous(lambda x, c: net(x, c).to(d), (B, 2, *RES), S, noises=tape, mode="ddim", ddim_eta=ETA, objective="eps", device=d)
    e = (got.to(d) - want).abs().flatten().double()
    print(f"ddim_eta1 |hip - oracle with the sampler scalars on the {where}|: q99 {torch.quantile(e, 0.99).item():.2e} rms {e.pow(2).mean().sqrt().item():.2e} max {e.max().item():.2e}", flush=True)
