#!/usr/bin/env python3
"""GPU-box fuzz: random U-Net geometries / batch sizes / precision modes against the float64 oracle on the device.

The engine picks a kernel family per layer from the shape (conv_f16x2 with >= 128 tiles, bf16x3 32- / 64-channel tiles, the
fp32-MFMA kernel for odd shapes, proj_f16x2 or fp32-MFMA 1x1, fused or streaming GroupNorm statistics, ...): every switch
point is a place where a fallback can be wrong and the fixed test configurations only visit a few.  Each case: setup_model on a
synthetic checkpoint of that geometry, one forward at a random batch <= max_batch, max |hip - fp64| (bar 2e-5 in the parity
modes), a repeated forward must be bit-identical, and a 2-step sample() must be finite."""
import os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from r2dm_amd._lib import R2DMError
from oracle import r2dm_oracle as O
dev = "cuda"
rnd = random.Random(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("CASES", "24"))
worst = 0.0; fails = 0
for case in range(N):
    H = rnd.choice([8, 16, 16, 32, 64]); W = rnd.choice([64, 128, 256, 512, 1024])
    if H * W > 64 * 1024: W = 64 * 1024 // H
    base = rnd.choice([16, 32, 48, 64, 64, 96, 128]); groups = rnd.choice([g for g in (2, 4, 8) if base % g == 0])
    mult = rnd.choice([(1, 2, 4, 8), (1, 2, 4, 8), (1, 1, 2, 4), (1, 2, 2, 4), (2, 2, 4, 4), (1, 2, 3, 4)])
    nres = rnd.choice([(3, 3, 3, 3), (1, 2, 2, 1), (2, 2, 2, 2), (1, 1, 1, 1)])
    heads = rnd.choice([h for h in (1, 2, 4, 8, 16) if (base * mult[3]) % h == 0 and (base * mult[2]) % h == 0 and (base * mult[3]) // h in (32, 64) and (base * mult[2]) // h in (32, 64)] or [0])
    enc = "fourier_features"  # (the oracle evaluates this encoding; the others are pinned by golden vectors)
    maxb = rnd.choice([1, 2, 3, 4, 8, 16]); B = rnd.randint(1, maxb); prec = rnd.choice(["fp32", "fp32", "fp32-bf16x3", "fp16"])
    kw = dict(resolution=(H, W), base_channels=base, gn_num_groups=groups, channel_multiplier=mult, num_residual_blocks=nres, coords_encoding=enc)
    if heads: kw["attn_num_heads"] = heads
    tag = f"case {case}: {H}x{W} base {base} mult {mult} res {nres} groups {groups} heads {heads} enc {enc} max_batch {maxb} batch {B} {prec}"
    try:
        ck = synthetic.synthetic_checkpoint(seed=case, **kw)
        ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=maxb, precision=prec)
    except (R2DMError, ValueError, AssertionError, TypeError) as e:
        print(tag, "-> rejected at set-up:", str(e)[:100], flush=True); continue
    g = torch.Generator(device=dev).manual_seed(case)
    x = torch.randn(B, 2, H, W, device=dev, generator=g); c = (torch.rand(B, device=dev, generator=g) - 0.5) * 20
    try:
        y = ddpm.model(x, c)
    except R2DMError as e:
        print(tag, "-> rejected at the first forward:", str(e)[:120], flush=True); continue
    sd = {k: v.double().to(dev) for k, v in O.strip_prefix(ck["ema_weights"]).items()}
    cfg = O.UNetConfig(resolution=(H, W), base_channels=base, channel_multiplier=mult, num_residual_blocks=nres, gn_num_groups=groups,
                       attn_num_heads=heads or 8)
    try:
        ref = O.unet_forward(sd, cfg, x.double(), c.double())
    except Exception as e:
        print(tag, "-> oracle cannot evaluate this geometry:", repr(e)[:100], flush=True); continue
    err = (y.double() - ref).abs().max().item(); scale = ref.abs().max().item()
    same = torch.equal(y, ddpm.model(x, c))
    s = ddpm.sample(batch_size=B, num_steps=2, progress=False, rng=r2dm_amd.setup_rng(list(range(B)), dev))
    bar = 2e-5 if prec != "fp16" else 2e-2
    ok = err < bar * max(1.0, scale) and same and bool(torch.isfinite(s).all())
    worst = max(worst, err if prec != "fp16" else 0.0); fails += (not ok)
    print(tag, f"-> max|hip - fp64| {err:.2e} (|ref| <= {scale:.2f}) repeat-identical {same} {'OK' if ok else 'FAIL'}", flush=True)
print(f"{N} cases, {fails} failures, worst parity-mode error {worst:.2e}")
