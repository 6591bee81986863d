#!/usr/bin/env python3
"""GPU-box probe (temporary build with s_memrealtime in prof slots 4/5): launch ramp / drain of one conv launch."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
B = int(os.environ.get("B", "8"))
dev = "cuda"
prof = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
os.environ["R2DM_CONV_PROF_PTR"] = str(prof.data_ptr())
from r2dm_amd import _lib
from bench_conv_shapes import SHAPES
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
for n in os.environ.get("SHAPES", "L1_64_64,L2_128_128,L3_256_256,L4_512_512").split(","):
    cin, cout, h, w, k, pro, res = SHAPES[n]
    x = torch.randn(B, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin*k*k)
    bias = torch.randn(cout, device=dev); aff = torch.rand(B, cin, 2, device=dev) + 0.5 if pro else None
    r = torch.randn(B, cout, h, w, device=dev) if res else None; sc = torch.tensor([0.7071], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device=dev); y = torch.empty(B, cout, h, w, device=dev)
    def go():
        _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro, _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
    for _ in range(5): go()
    prof.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); first = prof[0].clone()
    for _ in range(40): go()
    e1.record(); torch.cuda.synchronize()
    ticks_per_us = float(prof[0, 5] - first[4]) / (e0.elapsed_time(e1) * 1e3)
    prof.zero_(); go(); torch.cuda.synchronize()
    p = prof.cpu(); nb = int((p[:, 3] > 0).sum()); p = p[:nb]
    r0, r1 = p[:, 4].double() / ticks_per_us, p[:, 5].double() / ticks_per_us
    t0 = r0.min(); span = r1.max() - t0
    hw = p[:, 7] & 0xffffffff; xcc = (p[:, 7] >> 32) & 0xf
    key = (xcc * 8 + ((hw >> 13) & 7)) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    cu_first = torch.stack([r0[key == kk].min() for kk in key.unique()]) - t0
    cu_last = torch.stack([r1[key == kk].max() for kk in key.unique()]) - t0
    busy = sum(float(r1[key == kk].max() - r0[key == kk].min()) for kk in key.unique()) / len(key.unique())
    cyc = (p[:, 3] - p[:, 0]).double(); dur = (r1 - r0)
    print(f"{n}: span {span:.1f} us; CU first-start: median {cu_first.median():.1f} max {cu_first.max():.1f} us; CU last-end before span end: "
          f"median {(span - cu_last).median():.1f} max {(span - cu_last).max():.1f} us; mean CU busy {busy:.1f} us ({busy / span * 100:.1f}%); "
          f"realtime counter {ticks_per_us:.2f} ticks/us; shader clock {float((cyc / dur).median()) / 1e3:.3f} GHz")
