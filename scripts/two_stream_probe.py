#!/usr/bin/env python3
"""GPU-box probe (round 4): one sampler at batch 8 vs TWO samplers at batch 4 on two HIP streams (their persistent convolution
launches interleave at block granularity: one's prologue / tile-end burst / dependent-launch gap under the other's MFMA phase)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2dm_amd
from r2dm_amd import synthetic

dev = "cuda"
S = int(os.environ.get("STEPS", "48"))
ck = synthetic.synthetic_checkpoint(seed=0)
one, _, _ = r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=8)
halves = [r2dm_amd.setup_model(ck, device=dev, show_info=False, max_batch=4)[0] for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
steps = torch.linspace(1.0, 0.0, S + 1)


def run_one():
    x = torch.randn(8, 2, 64, 1024, device=dev)
    with one.model.deferred_range_check():
        for i in range(S):
            x = one.p_step(x, steps[i].repeat(8), steps[i + 1].repeat(8))
    return x


def run_two():
    xs = [torch.randn(4, 2, 64, 1024, device=dev) for _ in range(2)]
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    import contextlib
    with contextlib.ExitStack() as es:
        for h in halves:
            es.enter_context(h.model.deferred_range_check())
        for i in range(S):
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    xs[k] = halves[k].p_step(xs[k], steps[i].repeat(4), steps[i + 1].repeat(4))
        for s in streams:
            cur.wait_stream(s)
        torch.cuda.synchronize()
    return xs


for name, fn in (("one sampler, batch 8", run_one), ("two samplers, batch 4 + 4, two streams", run_two)) * 2:
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    fn()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{name}: {dt / S * 1e3:.3f} ms per step of 8 images", flush=True)
