#!/usr/bin/env python3
"""GPU-box probe: where (XCD/SE/CU) and when every block of one conv launch ran; are co-resident blocks in lockstep?"""
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
n = os.environ.get("SHAPE", "L1_64_64")
cin, cout, h, w, k, pro, res = SHAPES[n]
x = torch.randn(B, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin*k*k)
bias = torch.randn(cout, device=dev); aff = torch.rand(B, cin, 2, device=dev) + 0.5 if pro else None
r = torch.randn(B, cout, h, w, device=dev) if res else None; sc = torch.tensor([0.7071], device=dev) if res else None
packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device=dev); y = torch.empty(B, cout, h, w, device=dev)
for _ in range(3):
    prof.zero_()
    _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro, _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
    torch.cuda.synchronize()
# sustained clock: REPS back-to-back launches, wall time per launch vs the per-CU cycle span of the last one
REPS = int(os.environ.get("REPS", "300"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro, _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
e1.record(); torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) * 1e3 / REPS
p = prof.cpu()
nb = int((p[:, 3] > 0).sum()); p = p[:nb]
hw = p[:, 7] & 0xffffffff; xcc = (p[:, 7] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (xcc * 8 + se) * 32 + sh * 16 + cu
t00 = int(p[:, 0].min())
print(f"{n}: {nb} blocks on {len(key.unique())} CUs; kernel span {(int(p[:,3].max()) - t00)} cycles")
import collections
by = collections.defaultdict(list)
for i in range(nb):
    by[int(key[i])].append((int(p[i, 0]) - t00, int(p[i, 1]) - t00, int(p[i, 2]) - t00, int(p[i, 3]) - t00, i))
cnt = collections.Counter(len(v) for v in by.values()); print("blocks per CU histogram:", sorted(cnt.items()))
for kk in sorted(by)[:3] + sorted(by)[100:102]:
    print(f"CU key {kk} (xcc {kk // 256} se {(kk // 32) % 8} cu {kk % 32}):")
    for (a, b_, c, d, i) in sorted(by[kk]):
        print(f"   blk {i:5d}  start {a:8d}  main {b_:8d}  epi {c:8d}  end {d:8d}   (pro {b_-a:6d} main {c-b_:7d} epi {d-c:6d})")
# fraction of time with all resident blocks of a CU outside the main loop
idle = tot = 0
for kk, v in by.items():
    ev = []
    for (a, b_, c, d, i) in v: ev += [(b_, 1), (c, -1)]
    ev.sort(); cur = 0; start = min(a for (a, _, _, _, _) in v); last = start; end = max(d for (_, _, _, d, _) in v)
    for t, dlt in ev:
        if cur == 0: idle += t - last
        cur += dlt; last = t
    idle += end - last; tot += end - start
spans = [max(d for (_, _, _, d, _) in v) - min(a for (a, _, _, _, _) in v) for v in by.values()]
import statistics
print(f"sustained: {wall_us:.1f} us per launch (includes pack kernel + launch gap); per-CU busy span median {statistics.median(spans):.0f} max {max(spans)} cycles "
      f"-> shader clock >= {max(spans) / wall_us / 1e3:.2f} GHz if the span filled the launch")
print(f"CU-time with NO block in its main loop: {idle / tot * 100:.1f}% of the per-CU busy span")
