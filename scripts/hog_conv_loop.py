#!/usr/bin/env python3
"""GPU-box probe, neighbour role: one convolution shape of the tree in the CURRENT DIRECTORY in a loop for SECS seconds
(scripts/stress_cross.py, scripts/jobs/j77.sh).  SHAPE = cin,cout,h,w,ksize,batch (default: out_conv, 64 -> 2 at 64x1024)."""
import math, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from r2dm_amd import _lib
cin, cout, h, w, k, B = (int(v) for v in os.environ.get("SHAPE", "64,2,64,1024,3,8").split(","))
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k); b = torch.randn(cout, device="cuda")
packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device="cuda"); y = torch.empty(B, cout, h, w, device="cuda")
t0 = time.time(); n = 0
torch_only = os.environ.get("HOG_TORCH_ONLY") == "1"  # (no kernel of this library: element-wise torch kernels on a small tensor)
small = torch.randn(1 << 16, device="cuda")
while time.time() - t0 < float(os.environ.get("SECS", "120")):
    for _ in range(50):
        if torch_only:
            small.mul_(1.0001).add_(1e-3)
            continue
        _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), b.data_ptr(), packed.data_ptr(), None, 0, None, None, y.data_ptr(), B, cin, cout, h, w, k, st))
    torch.cuda.synchronize(); n += 50
    if os.environ.get("READY_FILE"): open(os.environ["READY_FILE"], "w").close()
print("hog_conv_loop:", n, "launches", flush=True)
