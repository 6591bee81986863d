#!/usr/bin/env python3
"""GPU-box probe: does replaying the U-Net forward from a captured HIP graph beat enqueueing its ~140 kernels one by one?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "8"))
ck = synthetic.synthetic_checkpoint(seed=0, resolution=(64, 1024))
ddpm, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=B)
net = ddpm.model
x = torch.randn(B, 2, 64, 1024, device=dev); c = torch.full((B,), -3.0, device=dev)
def bench(fn, n=100):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with net.deferred_range_check():
    y_ref = net(x, c).clone()
    t_eager = bench(lambda: net(x, c))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): net(x, c)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            y = net(x, c)
        g.replay(); torch.cuda.synchronize()
        ok = torch.equal(y, y_ref)
        t_graph = bench(g.replay)
        print(f"batch {B}: eager {t_eager:.3f} ms/forward, graph replay {t_graph:.3f} ms/forward, identical output: {ok}")
    except Exception as e:
        print(f"batch {B}: eager {t_eager:.3f} ms/forward; graph capture failed: {type(e).__name__}: {e}")
