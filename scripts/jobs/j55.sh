#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for lib in $R/r2dm_amd/libr2dm_hip.so $R/build_probe/lib_k812.so; do
echo "== $lib"
R2DM_HIP_LIB=$lib python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, math, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import hipops as H
from conftest import rnd
F = torch.nn.functional
cin, cout, h, w, B = 64, 64, 16, 256, 2
x, wt, b = rnd(1, B, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout) * 0
def ref_of(xx, ww): return F.conv2d(F.pad(F.pad(xx, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1)), ww, b)
# one input pixel column class (col % 4 == e) and one chunk at a time, centre tap only: y = w_c x restricted
for k in range(4):
  for e in range(4):
    xx = torch.zeros_like(x); xx[:, 16*k:16*k+16, :, e::4] = x[:, 16*k:16*k+16, :, e::4]
    ww = torch.zeros_like(wt); ww[:, :, 1, 1] = wt[:, :, 1, 1]
    y = H.conv2d_ring(xx.cuda(), ww.cuda(), b.cuda()).cpu()
    r = ref_of(xx, ww)
    err = (y - r).abs()
    bad = err > 1e-3
    cols = sorted(set((bad.nonzero()[:, 3] % 4).tolist()))
    print(f"chunk {k} quad pixel {e}: max err {err.max().item():.3e} bad {int(bad.sum())} of {bad.numel()}  bad output col%4 in {cols}  max|y| {y.abs().max().item():.3e}")
PY
done
