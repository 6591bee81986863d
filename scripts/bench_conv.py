#!/usr/bin/env python3
"""GPU-box microbenchmark of the conv kernel on the network's layer shapes (HIP-event timed, no syncs inside)."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2dm_amd import _lib

B = int(os.environ.get("B", "8"))
SHAPES = {  # name: (cin, cout, h, w, k, prologue, residual)
    "L1_64_64": (64, 64, 64, 1024, 3, 2, True),
    "L1_128_64": (128, 64, 64, 1024, 3, 2, False),
    "L1_64_128": (64, 128, 64, 1024, 3, 0, False),
    "L2_128_128": (128, 128, 32, 512, 3, 2, True),
    "L3_256_256": (256, 256, 16, 256, 3, 2, True),
    "L4_512_512": (512, 512, 8, 128, 3, 2, True),
    "L4_256_256": (256, 256, 8, 128, 3, 2, True),
    "L1_34_64": (34, 64, 64, 1024, 3, 0, False),
    "L1_64_2": (64, 2, 64, 1024, 3, 0, False),
    "L1_128_64_1x1": (128, 64, 64, 1024, 1, 0, False),
}
names = os.environ.get("SHAPES", ",".join(SHAPES)).split(",")
iters = int(os.environ.get("ITERS", "10"))
dev = "cuda"
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for n in names:
    cin, cout, h, w, k, pro, res = SHAPES[n]
    x = torch.randn(B, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, device=dev)
    aff = torch.rand(B, cin, 2, device=dev) + 0.5 if pro else None
    r = torch.randn(B, cout, h, w, device=dev) if res else None
    sc = torch.tensor([0.7071], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device=dev)
    y = torch.empty(B, cout, h, w, device=dev)
    call = lambda: _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro,
                                                _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    gf = 2.0 * B * cout * cin * k * k * h * w / 1e9
    ms = e0.elapsed_time(e1) / iters
    print(f"{n:16s} {gf:7.2f} GF  {ms*1e3:8.1f} us (conv + ~6us pack)  {gf/ms:7.1f} TF/s  {gf/ms/157.3*100:5.1f}%", flush=True)
