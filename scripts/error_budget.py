#!/usr/bin/env python3
"""GPU-box diagnostic: per-block error budget of the HIP U-Net against the fp64 oracle, next to the reference's own
fp32 arithmetic (the oracle in fp32 on the host CPU = torch/oneDNN, what tests/golden pins).

The fp64 oracle runs once (torch ops on the GPU) and records the input/output of every residual block, attention
block, resampler and free-standing convolution.  Each unit is then re-evaluated TEACHER-FORCED (its fp64 input rounded
to fp32) three ways: the HIP kernels through the single-kernel C ABI, the fp32 oracle on the CPU, and -- for the
convolutions inside residual blocks -- split into "prologue" (GroupNorm affine + SiLU) and "bare convolution" so the
source of a gap can be named.  Errors are relative to the rms of the unit's fp64 output.

    python scripts/error_budget.py            # 64x1024, batch 1, cond 0
    RES=16,128 COND=7 python scripts/error_budget.py
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops  # noqa: E402
from oracle import r2dm_oracle as O  # noqa: E402
from r2dm_amd import synthetic  # noqa: E402

DEV = "cuda"
res = tuple(int(v) for v in os.environ.get("RES", "64,1024").split(","))
cond_v = float(os.environ.get("COND", "0"))
torch.set_num_threads(min(32, os.cpu_count() or 1))

ck = synthetic.synthetic_checkpoint(seed=0, resolution=res)
sd32 = O.strip_prefix(ck["ema_weights"])
sd64 = {k: v.double().to(DEV) for k, v in sd32.items()}
sdg = {k: v.to(DEV) for k, v in sd32.items()}
cfg = O.UNetConfig(resolution=res)
G, EPS = cfg.gn_num_groups, cfg.gn_eps

records = []  # (kind, prefix, inputs..., out)
orig = {n: getattr(O, n) for n in ("residual_block", "self_attention_block", "fir_down2", "fir_up2", "conv_ring")}
depth = {"n": 0}


def wrap(kind):
    fn = orig[kind]

    def w(*a):
        depth["n"] += 1
        out = fn(*a)
        depth["n"] -= 1
        if depth["n"] == 0 or kind in ("residual_block", "self_attention_block"):
            records.append((kind, a, out))
        return out

    return w


for n in orig:
    setattr(O, n, wrap(n))
g = torch.Generator().manual_seed(1)
x = torch.randn(1, 2, *res, generator=g)
cond = torch.full((1,), cond_v)
y64 = O.unet_forward(sd64, cfg, x.double().to(DEV), cond.double().to(DEV))
for n, f in orig.items():
    setattr(O, n, f)


def err(a, ref):
    d = (a.double().to(ref.device) - ref)
    s = ref.pow(2).mean().sqrt().item()
    return d.pow(2).mean().sqrt().item() / s, d.abs().max().item() / s, d.mean().item() / s


def fmt(e):
    return f"rms {e[0]:.2e} max {e[1]:.2e} mean {e[2]:+.1e}"


def hip_residual(p, xin, temb):
    x32 = xin.float().contiguous()
    aff1, _ = hipops.group_norm_affine(x32, G, EPS, gamma=sdg[p + "norm1.weight"], beta=sdg[p + "norm1.bias"])
    t1 = hipops.conv2d_ring(x32, sdg[p + "conv1.weight"], sdg[p + "conv1.bias"], aff=aff1, prologue=2)
    ada = F.linear(O.silu(temb), sd64[p + "norm2.proj.1.weight"], sd64[p + "norm2.proj.1.bias"]).float().contiguous()
    aff2, _ = hipops.group_norm_affine(t1, G, EPS, ada=ada)
    if (p + "skip.weight") in sdg:
        r = hipops.conv2d_ring(x32, sdg[p + "skip.weight"], sdg[p + "skip.bias"])
    else:
        r = x32
    return hipops.conv2d_ring(t1, sdg[p + "conv2.weight"], sdg[p + "conv2.bias"], aff=aff2, prologue=2, residual=r,
                              scale=O.INV_SQRT2)


tot = {"hip": 0.0, "cpu": 0.0}
print(f"# resolution {res}, cond {cond_v}; errors relative to the rms of each unit's fp64 output")
print(f"{'unit':44s} {'HIP (teacher-forced)':44s} {'CPU fp32 oracle (teacher-forced)':44s}")
for kind, a, out in records:
    if kind == "residual_block":
        _, p, _, xin, temb = a
        eh = err(hip_residual(p, xin, temb), out)
        ec = err(O.residual_block(sd32, p, cfg, xin.float().cpu(), temb.float().cpu()), out)
        name = p + f" {tuple(xin.shape[1:])}"
        # split conv1: prologue vs bare convolution
        h64 = O.silu(O.group_norm(xin, G, EPS, sd64[p + "norm1.weight"], sd64[p + "norm1.bias"]))
        c64 = O.conv_ring(h64, sd64[p + "conv1.weight"], sd64[p + "conv1.bias"])
        h32 = h64.float().contiguous()
        bare_h = err(hipops.conv2d_ring(h32, sdg[p + "conv1.weight"], sdg[p + "conv1.bias"]), c64)
        bare_c = err(O.conv_ring(h32.cpu(), sd32[p + "conv1.weight"], sd32[p + "conv1.bias"]), c64)
        x32 = xin.float().contiguous()
        aff1, _ = hipops.group_norm_affine(x32, G, EPS, gamma=sdg[p + "norm1.weight"], beta=sdg[p + "norm1.bias"])
        pro_h = err(hipops.affine_act(x32, aff1, True), h64)
        pro_c = err(O.silu(O.group_norm(x32.cpu(), G, EPS, sd32[p + "norm1.weight"], sd32[p + "norm1.bias"])), h64)
        fused_h = err(hipops.conv2d_ring(x32, sdg[p + "conv1.weight"], sdg[p + "conv1.bias"], aff=aff1, prologue=2), c64)
        print(f"{name:44s} {fmt(eh):44s} {fmt(ec):44s}")
        print(f"{'   conv1 bare (input = fp64 silu(gn) rounded)':44s} {fmt(bare_h):44s} {fmt(bare_c):44s}")
        print(f"{'   conv1 prologue alone gn+silu':44s} {fmt(pro_h):44s} {fmt(pro_c):44s}")
        print(f"{'   conv1 fused gn+silu+conv':44s} {fmt(fused_h):44s}")
    elif kind == "self_attention_block":
        _, p, _, xin = a
        B, C, H, W = xin.shape
        x32 = xin.float().contiguous()
        aff, _ = hipops.group_norm_affine(x32, G, EPS, gamma=sdg[p + "norm.weight"], beta=sdg[p + "norm.bias"])
        qkv = hipops.conv2d_ring(x32, sdg[p + "attn.in_proj_weight"].view(3 * C, C, 1, 1).contiguous(), sdg[p + "attn.in_proj_bias"],
                                 aff=aff, prologue=1)
        o = hipops.attention(qkv.view(B, 3 * C, H * W), cfg.attn_num_heads).view(B, C, H, W)
        yh = hipops.conv2d_ring(o, sdg[p + "attn.out_proj.weight"].view(C, C, 1, 1).contiguous(), sdg[p + "attn.out_proj.bias"],
                                residual=x32, scale=O.INV_SQRT2)
        eh = err(yh, out)
        ec = err(O.self_attention_block(sd32, p, cfg, xin.float().cpu()), out)
        print(f"{p + f' {tuple(xin.shape[1:])}':44s} {fmt(eh):44s} {fmt(ec):44s}")
    elif kind in ("fir_down2", "fir_up2"):
        (xin,) = a
        f = hipops.fir_down2 if kind == "fir_down2" else hipops.fir_up2
        eh = err(f(xin.float().contiguous()), out)
        ec = err(orig[kind](xin.float().cpu()), out)
        print(f"{kind + f' {tuple(xin.shape[1:])}':44s} {fmt(eh):44s} {fmt(ec):44s}")
    else:  # free-standing convolution (in_conv, down / up convs, out_conv)
        xin, w, b = a
        eh = err(hipops.conv2d_ring(xin.float().contiguous(), w.float().contiguous(), b.float().contiguous()), out)
        ec = err(F.conv2d(O.ring_pad(xin.float().cpu(), w.shape[-1] // 2), w.float().cpu(), b.float().cpu()), out)
        print(f"{f'conv {tuple(w.shape)} @ {tuple(xin.shape[2:])}':44s} {fmt(eh):44s} {fmt(ec):44s}")
    tot["hip"] += eh[0] ** 2
    tot["cpu"] += ec[0] ** 2
print(f"root-sum-square of per-unit relative rms errors: HIP {tot['hip'] ** 0.5:.2e}   CPU fp32 {tot['cpu'] ** 0.5:.2e}")

# whole network, for scale
import r2dm_amd  # noqa: E402

ddpm, _, _ = r2dm_amd.setup_model(ck, device=DEV, show_info=False, max_batch=1)
yh = ddpm.model(x.to(DEV), cond.to(DEV))
yc = O.unet_forward(sd32, cfg, x, cond)
s = y64.pow(2).mean().sqrt().item()
print(f"whole U-Net (abs; output rms {s:.2f}): HIP-fp64 rms {(yh.double() - y64).pow(2).mean().sqrt().item():.2e} "
      f"max {(yh.double() - y64).abs().max().item():.2e} | CPU32-fp64 rms {(yc.double().to(DEV) - y64).pow(2).mean().sqrt().item():.2e} "
      f"max {(yc.double().to(DEV) - y64).abs().max().item():.2e}")
