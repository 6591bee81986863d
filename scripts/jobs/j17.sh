#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j17; mkdir -p $O; rm -f $O/cmp.log
cd $R
python scripts/dist_forward_probe.py /tmp/one > $O/one.log 2>&1
for rep in 1 2 3 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 scripts/dist_forward_probe.py /tmp/two > $O/two.log 2>&1
  python - <<'PY' 2>&1 | tee -a $O/cmp.log
import torch
one=torch.load('/tmp/one_rank0.pt'); t0=torch.load('/tmp/two_rank0.pt')
names=['fwd0','fwd1','fwd2','z0','sample1','sample2','xT','cond','coef','noise','x1','pred0','pred1','pred2','pred3']
print(' '.join(f"{n}:{(t0[i]-one[i]).abs().max().item():.2e}" for i,n in enumerate(names)))
p=[t0[i] for i in range(11,15)]
print('   rank0 preds vs own pred0:', [f"{(q-p[0]).abs().max().item():.2e}" for q in p], ' single preds vs own:', [f"{(one[i]-one[11]).abs().max().item():.2e}" for i in range(11,15)])
PY
done
