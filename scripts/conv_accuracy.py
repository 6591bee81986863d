#!/usr/bin/env python3
"""GPU-box diagnostic: error of the conv kernel (current R2DM_CONV_ALGO) against an fp64 CPU convolution, next to the
error of the same convolution done by torch in fp32 on the CPU (the oracle's arithmetic)."""
import os, sys, math
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hipops
torch.manual_seed(0)
print("R2DM_CONV_ALGO =", os.environ.get("R2DM_CONV_ALGO", "(default)"))
for name, (cin, cout, h, w, pro) in {"L1 64->64 @16x256": (64, 64, 16, 256, 2), "L1 64->64 no-prologue": (64, 64, 16, 256, 0), "L2 128->128 @16x128": (128, 128, 16, 128, 2),
                                      "L3 256->256 @8x128": (256, 256, 8, 128, 2), "L4 512->512 @8x128": (512, 512, 8, 128, 2),
                                      "L4 512->512 no-prologue": (512, 512, 8, 128, 0)}.items():
    B = 2
    x = torch.randn(B, cin, h, w); wt = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9); bias = torch.randn(cout) * 0.1
    aff = torch.stack([torch.rand(B, cin) + 0.5, torch.randn(B, cin) * 0.3], -1) if pro else None
    def ref(dt):
        xx = x.to(dt)
        if pro: xx = F.silu(xx * aff[..., 0].to(dt)[:, :, None, None] + aff[..., 1].to(dt)[:, :, None, None])
        xx = F.pad(F.pad(xx, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1))
        return F.conv2d(xx, wt.to(dt), bias.to(dt))
    r64, r32 = ref(torch.float64), ref(torch.float32)
    y = hipops.conv2d_ring(x.cuda(), wt.cuda(), bias.cuda(), aff=None if aff is None else aff.cuda().contiguous(), prologue=pro).cpu()
    e_h, e_c = (y.double() - r64), (r32.double() - r64)
    print(f"{name:28s} hip-fp64: max {e_h.abs().max():.2e} rms {e_h.pow(2).mean().sqrt():.2e} mean {e_h.mean():+.2e} | cpu32-fp64: max {e_c.abs().max():.2e} rms {e_c.pow(2).mean().sqrt():.2e}"
          f" | rel-slope {float((e_h * r64).sum() / (r64 * r64).sum()):+.2e}")
