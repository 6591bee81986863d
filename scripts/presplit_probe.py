#!/usr/bin/env python3
"""GPU-box probe (round 4): conv_f16x2 (64-channel tiles) with its stagers' transform vs the operand pre-pass (presplit.hip) +
DMA-only stagers, per layer shape: bit-equality of the outputs and HIP-event timings (pre-pass included), alternating."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2dm_amd import _lib

B = int(os.environ.get("B", "8"))
SHAPES = {  # name: (cin, cout, h, w, prologue, residual)
    "L4_512_512": (512, 512, 8, 128, 2, True),
    "U4_256_256": (256, 256, 8, 128, 2, True),
    "U4_512_256": (512, 256, 8, 128, 2, False),
    "L3_256_256": (256, 256, 16, 256, 2, True),
    "L3_256_512": (256, 512, 16, 256, 0, False),
    "U3_512_128": (512, 128, 16, 256, 2, False),
    "L2_128_128": (128, 128, 32, 512, 2, True),
    "L2_128_256": (128, 256, 32, 512, 0, False),
    "L1_64_64": (64, 64, 64, 1024, 2, True),
}
names = os.environ.get("SHAPES", ",".join(SHAPES)).split(",")
iters = int(os.environ.get("ITERS", "20"))
dev = "cuda"
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
os.environ["R2DM_F2_CO_TILE"] = "64"
PIECES = int(os.environ.get("PIECES", "2"))  # 2: the parity split (three products), 1: the fp16 bulk mode (one product)
_lib.check(L.r2dm_set_conv_pieces(None, PIECES))
print(f"# pieces = {PIECES}, batch {B}")
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

    def run(pre, y):
        os.environ["R2DM_F2_PRESPLIT"] = str(pre)
        _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro,
                                      _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, 3, st))

    ys = {c: torch.full((B, cout, h, w), float("nan"), device=dev) for c in (0, 1)}
    for c in ys:
        run(c, ys[c])
    torch.cuda.synchronize()
    same = torch.equal(ys[0], ys[1])
    dmax = (ys[0] - ys[1]).abs().max().item()
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
    print(f"{n:12s} bit-identical: {same} (max |d| {dmax:.2e}) | us (conv + weight pack [+ pre-pass]): stagers transform {min(t[0]):.1f}  pre-pass + DMA stagers {min(t[1]):.1f} | {gf:.1f} GF", flush=True)
