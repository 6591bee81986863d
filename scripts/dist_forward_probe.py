#!/usr/bin/env python3
"""GPU-box probe: U-Net forward of every rank of a (gloo, one GPU) multi-rank run vs a single process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
from r2dm_amd.distributed import broadcast_packed_weights
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
if world > 1:
    import torch.distributed as td
    td.init_process_group("gloo")
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device="cpu", show_info=False, max_batch=2)
ddpm.to(dev)
if os.environ.get("PROBE_SKIP_BCAST") != "1":
    broadcast_packed_weights(ddpm.model, dev, src=0)
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(2, 2, 64, 1024, device=dev, generator=g); c = torch.tensor([-15.0, -15.0], device=dev)
ys = [ddpm.model(x, c).cpu() for _ in range(3)]
seeds = [int(v) for v in os.environ.get("PROBE_SEEDS", "0,1").split(",")]
z0 = ddpm.randn(2, 2, 64, 1024, rng=r2dm_amd.setup_rng(seeds, dev), device=dev).cpu()
s1 = ddpm.sample(batch_size=2, num_steps=1, progress=False, rng=r2dm_amd.setup_rng(seeds, dev), return_all=True).cpu()
s2 = ddpm.sample(batch_size=2, num_steps=2, progress=False, rng=r2dm_amd.setup_rng(seeds, dev), return_all=True).cpu()
# the first sampling step spelled out
gens = r2dm_amd.setup_rng(seeds, dev)
xT = ddpm.randn(2, 2, 64, 1024, rng=gens, device=dev)
steps = torch.linspace(1.0, 0.0, 2)
cond, coef, mode_id = ddpm._coefficients(steps[:-1], steps[1:], "ddpm", 0.0)
cond_d = cond[:, None].expand(1, 2).contiguous().to(dev); coef_d = coef[:, None, :].expand(1, 2, 8).contiguous().to(dev)
preds = [ddpm.model(xT, cond_d[0]).cpu() for _ in range(4)]
noise = ddpm.randn_like(xT, rng=gens)
x1 = ddpm._posterior(xT, ddpm.model(xT, cond_d[0]), noise, coef_d[0], mode_id)
torch.save(ys + [z0, s1, s2, xT.cpu(), cond_d.cpu(), coef_d.cpu(), noise.cpu(), x1.cpu()] + preds, f"{sys.argv[1]}_rank{rank}.pt")
if world > 1:
    td.barrier(); td.destroy_process_group()
