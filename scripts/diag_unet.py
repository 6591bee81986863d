#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from oracle import r2dm_oracle as O
from r2dm_amd import synthetic
DEV="cuda"; res=tuple(int(v) for v in os.environ.get("RES","64,1024").split(","))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=res)
ddpm,_,_ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=1)
sd32 = O.strip_prefix(ck["ema_weights"]); sd64g = {k: v.double().to(DEV) for k,v in sd32.items()}
cfg = O.UNetConfig(resolution=res)
g = torch.Generator().manual_seed(1); x = torch.randn(1,2,*res, generator=g)
for c in (-15.0, 0.0, 7.0):
    cond = torch.full((1,), c)
    hip = ddpm.model(x.to(DEV), cond.to(DEV)).double().cpu()
    r64 = O.unet_forward(sd64g, cfg, x.double().to(DEV), cond.double().to(DEV)).cpu()
    d=(hip-r64).abs()
    r32 = O.unet_forward(sd32, cfg, x, cond).double()
    e=(r32-r64).abs()
    print("algo", os.environ.get("R2DM_CONV_ALGO","default"), f"cond {c}: hip-fp64 max {d.max():.2e} rms {d.pow(2).mean().sqrt():.2e} mean {(hip-r64).mean():+.2e} | cpu32-fp64 max {e.max():.2e} rms {e.pow(2).mean().sqrt():.2e} mean {(r32-r64).mean():+.2e}", flush=True)
