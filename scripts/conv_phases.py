#!/usr/bin/env python3
"""GPU-box probe: per-block phase timing (s_memtime) of the conv kernel for one layer shape."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(os.environ.get("B", "8"))
dev = "cuda"
prof = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
os.environ["R2DM_CONV_PROF_PTR"] = str(prof.data_ptr())
from r2dm_amd import _lib
from bench_conv_shapes import SHAPES
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
for n in os.environ.get("SHAPES", "L1_64_64").split(","):
    cin, cout, h, w, k, pro, res = SHAPES[n]
    x = torch.randn(B, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin*k*k)
    bias = torch.randn(cout, device=dev); aff = torch.rand(B, cin, 2, device=dev) + 0.5 if pro else None
    r = torch.randn(B, cout, h, w, device=dev) if res else None; sc = torch.tensor([0.7071], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device=dev); y = torch.empty(B, cout, h, w, device=dev)
    for _ in range(3):
        prof.zero_()
        _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro, _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
        torch.cuda.synchronize()
    p = prof.cpu()
    pi = p[p[:, 3] > 0]; p = pi.double()
    tot = (p[:, 3] - p[:, 0]).mean()
    ml = (p[:, 2] - p[:, 1]).mean()
    print(f"{n}: blocks {len(p)}  block {tot:.0f} cycles = prologue {(p[:,1]-p[:,0]).mean():.0f} + main loop {ml:.0f} + epilogue {(p[:,3]-p[:,2]).mean():.0f}")
    bar = (pi[:, 6] & 0xffffffff).double(); xf = (pi[:, 6] >> 32).double()
    for name, v in (("issue next loads", p[:, 4]), ("mfma + lds reads (+ staging / weight stores)", p[:, 5]), ("barrier", bar), ("chunk-boundary transform (bf16x3)", xf)):
        print(f"   {name:46s} {v.mean():9.0f} cycles ({v.mean() / ml * 100:5.1f}% of main loop)")
