#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j51; mkdir -p $O
cd $R
python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, math, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import hipops as H
from conftest import rnd
for (cin, cout, h, w, B) in [(64, 64, 16, 256, 2), (128, 64, 16, 256, 2), (128, 64, 16, 256, 1), (256, 64, 8, 128, 2), (128, 64, 64, 1024, 2)]:
    x, wt, b = rnd(1, B, cin, h, w), rnd(2, cout, cin, 3, 3) / math.sqrt(9 * cin), rnd(3, cout)
    y = H.conv2d_ring(x.cuda(), wt.cuda(), b.cuda()).cpu()
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(torch.nn.functional.pad(x, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1)), wt, b)
    bad = ~torch.isfinite(y) | ((y - ref).abs() > 1e-3)
    print((cin, cout, h, w, B), "bad", int(bad.sum()), "of", y.numel(), "max err", float((y - ref).abs().nan_to_num(9e9).max()))
    if bad.any():
        idx = bad.nonzero()
        print("   bad b", sorted(set(idx[:, 0].tolist())), "rows", sorted(set(idx[:, 2].tolist()))[:20], "cols range", int(idx[:, 3].min()), int(idx[:, 3].max()), "co", len(set(idx[:, 1].tolist())))
PY
