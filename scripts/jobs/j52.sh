#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for lib in $R/r2dm_amd/libr2dm_hip.so $R/build_probe/lib_seg3.so; do
echo "== $lib"
R2DM_HIP_LIB=$lib python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, math, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import hipops as H
from conftest import rnd
for (cin, cout, h, w, B) in [(64, 64, 16, 256, 2), (128, 64, 16, 256, 2), (256, 64, 8, 128, 2)]:
    x, wt, b = rnd(1, B, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    y = H.conv2d_ring(x.cuda(), wt.cuda(), b.cuda()).cpu()
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(torch.nn.functional.pad(x, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1)), wt, b)
    bad = ~torch.isfinite(y) | ((y - ref).abs() > 1e-3)
    print((cin, cout, h, w, B), "bad", int(bad.sum()), "of", y.numel(), "max err", float((y - ref).abs().nan_to_num(9e9).max()))
PY
done
