#!/usr/bin/env python3
"""GPU-box diagnostic (round 5): the FIR up-sampler alone, in a loop, on fixed inputs (the three shapes of the 64x1024 network at batch 2), while a second
process keeps the GPU busy -- does ITS OUTPUT change?  (The branch-free rewrite of fir_up2 fails the two-rank drop-in scenario 5 of 5: is the kernel itself the
victim, or does it only trigger a failure elsewhere?)  Library: R2DM_HIP_LIB or the default.  NEIGHBOUR = conv (scripts/hog_conv_loop.py) | sampler | none."""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H
from conftest import rnd
shapes = [(2, 256, 8, 128), (2, 128, 16, 256), (2, 64, 32, 512)]
xs = [rnd(90 + i, *s).cuda() for i, s in enumerate(shapes)]
ref = [H.fir_up2(x).clone() for x in xs]
nb = os.environ.get("NEIGHBOUR", "conv")
hog = None
if nb == "conv":
    ready = "/tmp/fir_soak_ready"
    if os.path.exists(ready): os.remove(ready)
    hog = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "hog_conv_loop.py")], env=dict(os.environ, SHAPE=os.environ.get("HOG_SHAPE", "64,64,64,1024,3,8"), SECS="40", READY_FILE=ready),  # (HOG_TORCH_ONLY=1: element-wise torch kernels only)
                           cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t0 = time.time()
    while not os.path.exists(ready) and time.time() - t0 < 60: time.sleep(0.5)
elif nb == "sampler":
    hog = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); import torch, r2dm_amd; from conftest import synthetic_ckpt\n"
                            "m, _, _ = r2dm_amd.setup_model(synthetic_ckpt(resolution=(64, 1024)), device='cuda', show_info=False, max_batch=2)\n"
                            "import time; t0 = time.time()\n"
                            "while time.time() - t0 < 40: m.sample(2, 2, progress=False); torch.cuda.synchronize()"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(12)
bad = [0, 0, 0]; shown = [0, 0, 0]; n = 0
t0 = time.time()
while time.time() - t0 < float(os.environ.get("SECS", "15")):
    for k, x in enumerate(xs):
        y = H.fir_up2(x)
        if not torch.equal(y, ref[k]):
            bad[k] += 1
            if shown[k] < 3:  # what the corruption looks like: where, how much, against what
                shown[k] += 1
                d = (y != ref[k])
                idx = d.nonzero()
                b_, c_, r_, w_ = (idx[:, j] for j in range(4))
                yv, rv = y[d], ref[k][d]
                print(f"  shape {tuple(x.shape)}: {int(d.sum())} of {d.numel()} elements differ; batch {sorted(set(b_.tolist()))} planes {int(c_.min())}..{int(c_.max())} ({len(set(c_.tolist()))} distinct) "
                      f"rows {sorted(set(r_.tolist()))[:12]} cols {int(w_.min())}..{int(w_.max())}; got zeros {int((yv == 0).sum())}, ref zeros {int((rv == 0).sum())}, max |diff| {float((yv - rv).abs().max()):.3g}, "
                      f"first (b, c, row, col) {idx[0].tolist()} got {float(yv[0]):.6g} want {float(rv[0]):.6g}", flush=True)
    n += 1
print(f"fir_up_soak: library {os.environ.get('R2DM_HIP_LIB', 'default')} neighbour {nb}: {n} rounds, outputs differing from the first per shape {bad}", flush=True)
if hog: hog.terminate(); hog.wait()
