#!/usr/bin/env python3
"""GPU-box probe: forwards of every precision mode (batch 2 and 8 at 64x1024) with the library R2DM_HIP_LIB selects, saved to OUT (first
run) or compared bit for bit with the file OUT of an earlier run (second run): is a rebuilt kernel bit-identical to its predecessor?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import r2dm_amd
from conftest import synthetic_ckpt, rnd
out = os.environ["OUT"]
res = {}
for batch in (2, 8):
    ddpm, _, _ = r2dm_amd.setup_model(synthetic_ckpt(), device="cuda", show_info=False, max_batch=batch)
    x, c = rnd(91, batch, 2, 64, 1024).to("cuda"), torch.linspace(-3.0, 1.0, batch, device="cuda")
    for prec in ("fp32", "fp32-bf16x3", "fp16"):
        ddpm.model.set_precision(prec)
        res[f"{batch}/{prec}"] = ddpm.model(x, c).cpu()
if os.path.exists(out):
    ref = torch.load(out)
    for k in res:
        d = (res[k].double() - ref[k].double()).abs().max().item()
        print(f"ab_bits {k}: bit-identical {torch.equal(res[k], ref[k])}  max |d| {d:.3e}", flush=True)
else:
    torch.save(res, out)
    print("ab_bits: saved", out, os.environ.get("R2DM_HIP_LIB"), flush=True)
