#!/usr/bin/env python3
"""GPU-box probe (round 4): conv_f16x2 with 64-channel tiles (two accumulators) vs 128-channel tiles (one accumulator) on the
network's layer shapes -- error of both against an fp64 convolution (one sample) and HIP-event timings, alternating."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2dm_amd import _lib

B = int(os.environ.get("B", "8"))
SHAPES = {  # name: (cin, cout, h, w, prologue, residual)
    "L1_64_128": (64, 128, 64, 1024, 0, False),
    "L2_128_128": (128, 128, 32, 512, 2, True),
    "L2_128_256": (128, 256, 32, 512, 0, False),
    "L3_256_256": (256, 256, 16, 256, 2, True),
    "L3_256_512": (256, 512, 16, 256, 0, False),
    "L4_512_512": (512, 512, 8, 128, 2, True),
    "U3_512_128": (512, 128, 16, 256, 2, False),
    "L1_64_64": (64, 64, 64, 1024, 2, True),
}
names = os.environ.get("SHAPES", ",".join(SHAPES)).split(",")
iters = int(os.environ.get("ITERS", "20"))
dev = "cuda"
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for n in names:
    cin, cout, h, w, pro, res = SHAPES[n]
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, cin, h, w, device=dev, generator=g)
    wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / math.sqrt(cin * 9)
    bias = torch.randn(cout, device=dev, generator=g)
    aff = torch.stack([torch.rand(B, cin, device=dev, generator=g) + 0.5, torch.randn(B, cin, device=dev, generator=g) * 0.3], -1).contiguous() if pro else None
    r = torch.randn(B, cout, h, w, device=dev, generator=g) if res else None
    sc = torch.tensor([0.70710678], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, 3, B, h, w), device=dev)

    def run(cot, y):
        os.environ["R2DM_F2_CO_TILE"] = str(cot)
        _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro,
                                      _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, 3, st))

    ys = {c: torch.empty(B, cout, h, w, device=dev) for c in (64, 128)}
    for c in ys:
        run(c, ys[c])
    torch.cuda.synchronize()
    bs = B - 1
    xa = x[bs:].double()
    if pro:
        xa = xa * aff[bs:, :, 0].double()[:, :, None, None] + aff[bs:, :, 1].double()[:, :, None, None]
    if pro == 2:
        xa = F.silu(xa)
    xp = F.pad(F.pad(xa, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1))
    ref = F.conv2d(xp, wt.double(), bias.double())
    y32 = F.conv2d(xp.float(), wt, bias).double()  # stock fp32 convolution of the same operands (yardstick)
    if res:
        ref = (r[bs:].double() + ref) * 0.70710678
        y32 = (r[bs:].double() + y32) * 0.70710678
    scale = ref.pow(2).mean().sqrt().item()
    line = f"{n:12s}"
    for c in ys:
        d = ys[c][bs:].double() - ref
        line += f" | {c:3d}: rel rms {d.pow(2).mean().sqrt().item() / scale:.2e} max {d.abs().max().item():.2e} bias {d.mean().item() / scale:+.1e}"
    d = y32 - ref
    line += f" | torch fp32: rel rms {d.pow(2).mean().sqrt().item() / scale:.2e} max {d.abs().max().item():.2e}"
    line += f" | 64 vs 128 equal: {torch.equal(ys[64], ys[128])}"
    # timings (the packing kernels run inside the call: ~6 us, the same for both)
    t = {}
    for rep in range(2):
        for c in ys:
            for _ in range(3):
                run(c, ys[c])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                run(c, ys[c])
            e1.record()
            torch.cuda.synchronize()
            t.setdefault(c, []).append(e0.elapsed_time(e1) / iters * 1e3)
    gf = 2.0 * B * cout * cin * 9 * h * w / 1e9
    line += " | us (conv + pack): " + " ".join(f"{c}: {min(v):.1f}" for c, v in t.items()) + f" | {gf:.1f} GF"
    print(line, flush=True)
