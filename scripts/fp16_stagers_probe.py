#!/usr/bin/env python3
"""GPU-box probe (round 4): the fp16 bulk mode's convolutions (pieces = 1) per layer shape: sha1 of the outputs (two libraries must agree bit for
bit: R2DM_HIP_LIB / R2DM_F2_STAGERS) and HIP-event timings."""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2dm_amd import _lib

B = int(os.environ.get("B", "8"))
SHAPES = {  # name: (cin, cout, h, w, prologue, residual)
    "L1_64_64": (64, 64, 64, 1024, 2, True),
    "L1_64_128": (64, 128, 64, 1024, 0, False),
    "L2_128_128": (128, 128, 32, 512, 2, True),
    "L3_256_256": (256, 256, 16, 256, 2, True),
    "L4_512_512": (512, 512, 8, 128, 2, True),
    "U4_256_256": (256, 256, 8, 128, 2, True),
    "U3_512_128": (512, 128, 16, 256, 1, False),
    "edge_64_64": (64, 64, 4, 64, 2, True),
}
dev = "cuda"
L = _lib.lib()
_lib.check(L.r2dm_set_conv_pieces(None, int(os.environ.get("PIECES", "1"))))
os.environ["R2DM_F2_CO_TILE"] = os.environ.get("COT", "64")
st = torch.cuda.current_stream().cuda_stream
iters = int(os.environ.get("ITERS", "20"))
for n, (cin, cout, h, w, pro, res) in SHAPES.items():
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, cin, h, w, device=dev, generator=g)
    wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / math.sqrt(cin * 9)
    bias = torch.randn(cout, device=dev, generator=g)
    aff = torch.stack([torch.rand(B, cin, device=dev, generator=g) + 0.5, torch.randn(B, cin, device=dev, generator=g) * 0.3], -1).contiguous() if pro else None
    r = torch.randn(B, cout, h, w, device=dev, generator=g) if res else None
    sc = torch.tensor([0.70710678], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, 3, B, h, w), device=dev)
    y = torch.full((B, cout, h, w), float("nan"), device=dev)
    call = lambda: _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro,
                                                _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, 3, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    sha = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"{n:12s} sha1 {sha} finite {bool(torch.isfinite(y).all())}  {e0.elapsed_time(e1) / iters * 1e3:8.1f} us (conv + weight pack)", flush=True)
