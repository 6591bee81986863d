#!/usr/bin/env python3
"""GPU-box probe: are the split-operand convolution kernels' results independent of what else runs on the GPU?

A second process keeps the GPU busy with (HOG=) "copy" (HBM-bound copies), "gemm" (bf16 matmul) or "conv" (the same HIP
convolution kernels); this process runs one convolution launch ITERS times and counts outputs that differ bitwise from the
first one (the kernels are deterministic when alone).  Wrong results under a memory-bound neighbour would mean a wait in
the kernel that only holds at uncontended latencies."""
import math, os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2dm_amd import _lib
from bench_conv_shapes import SHAPES

def conv_call(name, B=8, dev="cuda"):
    L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
    cin, cout, h, w, k, pro, res = SHAPES[name]
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, cin, h, w, device=dev, generator=g); wt = torch.randn(cout, cin, k, k, device=dev, generator=g) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, device=dev, generator=g); aff = torch.rand(B, cin, 2, device=dev, generator=g) + 0.5 if pro else None
    r = torch.randn(B, cout, h, w, device=dev, generator=g) if res else None; sc = torch.tensor([0.7071], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device=dev); y = torch.empty(B, cout, h, w, device=dev)
    def call():
        _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro, _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
        return y
    return call

if len(sys.argv) > 1 and sys.argv[1] == "hog":
    kind, secs = sys.argv[2], float(sys.argv[3])
    t0 = time.time()
    if kind == "copy":
        a = torch.empty(1 << 28, device="cuda"); b = torch.empty_like(a)
        while time.time() - t0 < secs:
            for _ in range(20): b.copy_(a)
            torch.cuda.synchronize()
    elif kind == "gemm":
        a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        while time.time() - t0 < secs:
            for _ in range(20): a @ a
            torch.cuda.synchronize()
    elif kind == "forward":
        import r2dm_amd
        from r2dm_amd import synthetic
        ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
        ddpm, _, _ = r2dm_amd.setup_model(ck, device=torch.device("cuda", 0), show_info=False, max_batch=2)
        xx = torch.randn(2, 2, 64, 1024, device="cuda"); cc = torch.full((2,), -3.0, device="cuda")
        with ddpm.model.deferred_range_check():
            while time.time() - t0 < secs:
                for _ in range(10): ddpm.model(xx, cc)
                torch.cuda.synchronize()
    else:
        c = conv_call("L1_64_64")
        while time.time() - t0 < secs:
            for _ in range(50): c()
            torch.cuda.synchronize()
    sys.exit(0)

iters = int(os.environ.get("ITERS", "300"))
import ctypes
L = _lib.lib(); L.r2dm_set_conv_pieces.argtypes = [ctypes.c_void_p, ctypes.c_int32]
for hog in os.environ.get("HOGS", "none,copy,gemm,conv").split(","):
    p = None
    if hog != "none":
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "hog", hog, "60"])
        time.sleep(30 if hog == "forward" else 8)  # let it start (import torch) and run
    for pieces in (2, 3):
        _lib.check(L.r2dm_set_conv_pieces(None, pieces))
        for name in os.environ.get("SHAPES", "L1_64_64,L3_256_256").split(","):
            c = conv_call(name)
            ref = c().clone(); torch.cuda.synchronize()
            bad = 0; worst = 0.0; t0 = time.time()
            for i in range(iters):
                y = c()
                if not torch.equal(y, ref):
                    bad += 1; worst = max(worst, (y - ref).abs().max().item())
                    if bad <= 3:  # where: (sample, 32-channel block, tile row of 4, 32-column segment) cells that differ
                        d = (y != ref)
                        Bq, Cq, Hq, Wq = d.shape
                        cells = d.view(Bq, Cq // 32, 32, Hq // 4, 4, Wq // 32, 32).any(dim=6).any(dim=4).any(dim=2).nonzero().tolist()
                        rows = d.view(Bq, Cq, Hq // 4, 4, Wq).any(dim=4).any(dim=1).any(dim=0).any(dim=0).tolist()
                        print(f"    mismatch {bad}: {int(d.sum())} elements in {len(cells)} cells (b, c/32, h/4, w/32), first {cells[:8]}; rows within the 4-row tile hit: {rows}; max |diff| {(y - ref).abs().max().item():.2e}", flush=True)
            torch.cuda.synchronize()
            print(f"neighbour={hog:5s} pieces={pieces} {name:12s}: {bad:3d} of {iters} launches differ from the first (max |diff| {worst:.2e}); {(time.time() - t0) / iters * 1e6:.0f} us/launch", flush=True)
    if p is not None:
        p.terminate(); p.wait()
