#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, math, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import hipops as H
from conftest import rnd
F = torch.nn.functional
cin, cout, h, w, B = 128, 64, 16, 256, 2
x, wt, b = rnd(1, B, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout) * 0
def ref_of(xx, ww): return F.conv2d(F.pad(F.pad(xx, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1)), ww, b)
for k in range(8):
    ww = torch.zeros_like(wt); ww[:, 16 * k:16 * k + 16] = wt[:, 16 * k:16 * k + 16]
    y = H.conv2d_ring(x.cuda(), ww.cuda(), b.cuda()).cpu()
    r = ref_of(x, ww)
    # does y match the reference computed with the activations of ANOTHER chunk?
    best = min(((y - ref_of(torch.roll(x, shifts=16 * s, dims=1), ww)).abs().max().item(), s) for s in range(-7, 8))
    print(f"only chunk {k} nonzero: max err {(y - r).abs().max().item():.3e}; best match when the input channels are rolled by {best[1]} chunks: {best[0]:.3e}")
PY
