#!/usr/bin/env python3
"""GPU-box probe: barrier / epilogue timeline of block 0 of conv_f16x2_kernel (library built with -DF2_PROF)."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(os.environ.get("B", "8"))
dev = "cuda"
prof = torch.zeros(4096, dtype=torch.int64, device=dev)
os.environ["R2DM_CONV_PROF_PTR"] = str(prof.data_ptr())
from r2dm_amd import _lib
from bench_conv_shapes import SHAPES
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
_lib.check(L.r2dm_set_conv_pieces(None, int(os.environ.get('PIECES', '2'))))  # 1: the fp16 bulk mode's kernels
NAMES = {1: "M arr#1", 2: "M arr#2", 3: "M arr#3", 4: "M lv#1", 5: "M lv#2", 6: "M lv#3", 7: "M epi begin", 8: "M epi end", 20: "M epi bias+res requested", 26: "M epi accumulators merged", 27: "M epi residual requested", 28: "M epi q0 turned", 21: "M epi quarter 0 done", 22: "M epi m=0 quarters done", 23: "M epi m=0 stats written", 24: "M epi m=1 quarters done", 25: "M epi m=1 stats written",
         30: "h start", 50: "h cursors set", 51: "h ad4 requested", 52: "h ring requested", 53: "h table requested", 54: "h chunk 0 requested", 31: "h prologue loads issued", 32: "h chunk 0 landed", 33: "h chunk 0 transformed", 34: "h arr P", 35: "h lv P", 36: "M arr P", 37: "M lv P", 60: "M slice begin", 61: "M slice operands landed", 62: "M slice quarter stored", 63: "M slice end", 40: "M tail begin", 41: "M tail q1", 42: "M tail q2", 43: "M tail q3",
         10: "h xf0 done", 11: "h arr#1", 12: "h lv#1", 13: "h xf1 done", 14: "h arr#2", 15: "h lv#2", 16: "h loads issued", 17: "h arr#3", 18: "h lv#3"}
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
    p = prof.cpu().tolist()
    r0, c0, r1, c1 = p[2048:2052]
    if r1 > r0: print(f"== {n}: block 0 ran {(r1 - r0) / 100:.1f} us (100 MHz counter), {c1 - c0} s_memtime ticks -> {(c1 - c0) / ((r1 - r0) / 100) / 1000:.3f} GHz")
    ev = []
    for g in (0, 1):
        for v in p[g*1024:(g+1)*1024]:
            if v: ev.append((v >> 8, g, v & 255))
    ev.sort()
    t0 = ev[0][0]
    print(f"== {n}: first event {t0 - c0} ticks after the block's start, last event {c1 - ev[-1][0]} ticks before its end")
    print(f"== {n}: {len(ev)} events; columns: s_memtime ticks since first event | wave | event | delta since the wave's previous event")
    last = {0: t0, 1: t0}
    for t, g, c in ev[:int(os.environ.get("MAXEV", "400"))]:
        print(f"{t - t0:9d}  {'M' if g == 0 else '        h'}  {str(NAMES.get(c, c)):14s} +{t - last[g]}")
        last[g] = t
