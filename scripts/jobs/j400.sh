#!/bin/bash
# round 6, first job: baseline of HEAD on this round's box + what the nine torch.randn launches of a step cost (side stream | main stream | none: timing only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j400; mkdir -p $O
cd $R
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in side main none; do
    case $m in side) E="";; main) E="R2DM_NOISE_STREAM=0";; none) E="R2DM_DEBUG_FIXED_NOISE=1";; esac
    env $E timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench noise=$m', round(j['ms_per_step'],3), round(j['value'],3), j['roofline']['frac'])"
  done
done | tee $O/ab_noise.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python $R/bench.py $A --steps 8 --warmup 2 --prewarm-s 0.5 > $O/kt.json 2> $O/kt.err
F=$(find $O -name "kt_kernel_trace.csv" | head -1)
python - "$F" > $O/overlap.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 4000 kernels: overlap of the randn launches with what else runs
rows = rows[-4000:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
tot = collections.Counter(); ov = collections.Counter()
for i, (s, e, n, q) in enumerate(iv):
    if "distribution" not in n: continue
    for j in range(max(0, i - 20), min(len(iv), i + 20)):
        if j == i: continue
        s2, e2, n2, q2 = iv[j]
        o = min(e, e2) - max(s, s2)
        if o > 0: ov[n2] += o
    tot[n] += e - s
print("randn total ns", dict(tot)); print("overlapped with:"); [print(f"  {k:42s} {v}") for k, v in ov.most_common(12)]
t0 = iv[0][0]; t1 = max(e for _, e, _, _ in iv); busy = sum(e - s for s, e, n, _ in iv if "distribution" not in n)
print("span ns", t1 - t0, "sum of non-randn kernel ns", busy)
PY
rm -f $F
cat $O/overlap.txt
