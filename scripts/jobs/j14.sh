#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j14; mkdir -p $O
cd $R
C="--steps 2 --warmup 1 --batch 2 --no-cpu-baseline --no-torch-baseline"
R2DM_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --dump-samples /tmp/two $C > $O/two.log 2>&1
python bench.py --gpus 1 --seed-base 0 --dump-samples /tmp/a0 $C > $O/a0.log 2>&1
python bench.py --gpus 1 --seed-base 0 --dump-samples /tmp/a0b $C > $O/a0b.log 2>&1
python bench.py --gpus 1 --seed-base 2 --dump-samples /tmp/a2 $C > $O/a2.log 2>&1
python - <<'PY' > $O/cmp.log 2>&1
import torch
t0=torch.load('/tmp/two/rank0.pt'); t1=torch.load('/tmp/two/rank1.pt'); a0=torch.load('/tmp/a0/rank0.pt'); a0b=torch.load('/tmp/a0b/rank0.pt'); a2=torch.load('/tmp/a2/rank0.pt')
def d(x,y): return (x['samples']-y['samples']).abs().max().item(), (x['samples']!=y['samples']).float().mean().item()
print('single vs single (same seeds):', d(a0,a0b))
print('rank0 vs single:', d(t0,a0), t0['seeds'], a0['seeds'])
print('rank1 vs single:', d(t1,a2), t1['seeds'], a2['seeds'])
PY
cat $O/cmp.log; tail -2 $O/two.log | cut -c1-300
