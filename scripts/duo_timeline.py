#!/usr/bin/env python3
"""GPU-box probe: phase timeline of block 0 of conv_bf16x3_duo_kernel (library built with -DDUO_PROF)."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(os.environ.get("B", "8"))
dev = "cuda"
prof = torch.zeros(4096, dtype=torch.int64, device=dev)
os.environ["R2DM_CONV_PROF_PTR"] = str(prof.data_ptr())
os.environ.setdefault("R2DM_DUO_MIN", "1")
from r2dm_amd import _lib
from bench_conv_shapes import SHAPES
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
NAMES = {19: "flip done", 20: "xf0 done", 21: "dma done", 22: "xf1+lgkm", 23: "dma done", 24: "frag_first", 25: "dma done", 30: "E:xf0 done", 31: "E:begin+ld01", 32: "E:dma", 33: "E:xf1+lgkm", 34: "E:fin00", 35: "E:ld10+fin01", 36: "E:ld11+stats0", 37: "E:dma", 38: "E:fin10", 39: "E:fin11", 40: "E:stats1", 41: "E:end", 42: "E:frag_first", 43: "E:dma", 44: "next_raw", 1: "compute", 2: "arrive B'", 8: "leave B'", 3: "between", 7: "between(END)", 9: "arr#1", 4: "lv#1", 10: "arr#2", 5: "lv#2", 11: "arr#3", 6: "lv#3"}
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
    ev = []
    for g in (0, 1):
        for v in p[g*1024:(g+1)*1024]:
            if v: ev.append((v >> 8, g, v & 255))
    ev.sort()
    t0 = ev[0][0]
    print(f"== {n}: {len(ev)} events; columns: cycles since first event | group | event | delta since the group's previous event")
    last = {0: t0, 1: t0}
    for t, g, c in ev:
        print(f"{t - t0:9d}  {'L' if g == 0 else '        F'}  {NAMES.get(c, c):14s} +{t - last[g]}")
        last[g] = t
