#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j16; mkdir -p $O
cd $R
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 scripts/dist_forward_probe.py /tmp/two > $O/two.log 2>&1
python scripts/dist_forward_probe.py /tmp/one > $O/one.log 2>&1
python - <<'PY' 2>&1 | tee $O/cmp.log
import torch
one=torch.load('/tmp/one_rank0.pt'); t0=torch.load('/tmp/two_rank0.pt'); t1=torch.load('/tmp/two_rank1.pt')
names=['fwd0','fwd1','fwd2','z0','sample1','sample2']
for name, t in (('rank0', t0), ('rank1', t1)):
    for i in range(6):
        d=(t[i]-one[i]).abs()
        print(name, names[i], 'max diff vs single', d.max().item(), 'frac differing', (d>0).float().mean().item())
s=t0[5]; o=one[5]
for k in range(s.shape[0]): print('rank0 sample2 step', k, (s[k]-o[k]).abs().max().item())
PY
