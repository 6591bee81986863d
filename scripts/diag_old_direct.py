#!/usr/bin/env python3
"""GPU-box probe (round 3, weak #1, last step): WHAT does the old LDS-tiled out_conv kernel (conv_direct_kernel<2> of the tree in
the CURRENT DIRECTORY, e.g. build_probe/bis_d5e0cd1) compute wrongly next to a neighbour process?

Integer-valued inputs and weights make every partial sum exact in fp32, so a wrong output pixel can be matched against
hypotheses: H1 one 8-channel chunk of the LDS halo tile was read while it still held the PREVIOUS chunk's channels (a read
that overtook the staging of its chunk), H2 ... the NEXT chunk's (a write that overtook the reads of the chunk before), H3 a
chunk's contribution is missing.  Prints the lane / row structure of the wrong pixels."""
import math, os, shlex, signal, subprocess, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from r2dm_amd import _lib
dev = "cuda"
B, cin, cout, H, W = 8, 64, 2, 64, 1024
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randint(-4, 5, (B, cin, H, W), device=dev, generator=g).float()
wt = torch.randint(-2, 3, (cout, cin, 3, 3), device=dev, generator=g).float()
b = torch.zeros(cout, device=dev)
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, 3, B, H, W), device=dev); y = torch.empty(B, cout, H, W, device=dev)
def call():
    _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), b.data_ptr(), packed.data_ptr(), None, 0, None, None, y.data_ptr(), B, cin, cout, H, W, 3, st))
    return y
def ring(xx, ww):  # reference: circular in W, zero in H
    return F.conv2d(F.pad(F.pad(xx, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1)), ww)
ref = ring(x, wt)
assert torch.equal(call(), ref), "alone, integer data: the kernel must be exact"
chunks = [ring(x[:, k * 8:(k + 1) * 8], wt[:, k * 8:(k + 1) * 8]) for k in range(8)]                      # C_k: chunk k's own contribution
prev = [ring(x[:, (k - 1) * 8:k * 8], wt[:, k * 8:(k + 1) * 8]) if k > 0 else None for k in range(8)]        # chunk k's weights on chunk k-1's pixels
nxt = [ring(x[:, (k + 1) * 8:(k + 2) * 8], wt[:, k * 8:(k + 1) * 8]) if k < 7 else None for k in range(8)]   # ... on chunk k+1's pixels
p = subprocess.Popen(os.environ["HOG_CMD"], shell=True, preexec_fn=os.setsid, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
time.sleep(float(os.environ.get("HOG_WAIT", "25")))
bad = 0; shown = 0; hyp = {"H1 stale previous chunk": 0, "H2 next chunk already there": 0, "H3 chunk missing": 0, "unexplained": 0}; lanes = {}
for it in range(int(os.environ.get("ITERS", "400"))):
    out = call().clone()
    d = out != ref
    if not d.any(): continue
    bad += 1
    idx = d.nonzero()
    for (bb, co, r, c) in idx[:2000].tolist():
        got, want = out[bb, co, r, c].item(), ref[bb, co, r, c].item()
        tag = "unexplained"
        for k in range(8):
            base = want - chunks[k][bb, co, r, c].item()
            if prev[k] is not None and got == base + prev[k][bb, co, r, c].item(): tag = "H1 stale previous chunk"; break
            if nxt[k] is not None and got == base + nxt[k][bb, co, r, c].item(): tag = "H2 next chunk already there"; break
            if got == base: tag = "H3 chunk missing"; break
        hyp[tag] += 1
        key = (r % 4, (c % 64) // 16)
        lanes[key] = lanes.get(key, 0) + 1
    if shown < 4:
        shown += 1
        rows = sorted(set(idx[:, 2].tolist())); cols = sorted(set((idx[:, 3] // 16 * 16).tolist()))
        print(f"  launch {it}: {len(idx)} wrong elements; samples {sorted(set(idx[:,0].tolist()))} channels {sorted(set(idx[:,1].tolist()))} rows {rows[:10]} 16-column groups starting at {cols[:10]}", flush=True)
print(f"{bad} launches with wrong pixels; hypotheses over the wrong elements: {hyp}; (row in the 4-row tile, 16-lane group of the wave) histogram: {dict(sorted(lanes.items()))}", flush=True)
os.killpg(os.getpgid(p.pid), signal.SIGTERM); p.wait()
