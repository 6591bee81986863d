#!/bin/bash
# round 6: why are the small fir_up2 launches slower than in the round-5 tree (11 / 17 us -> 18 / 30 us)?  Suspect: the per-site range slots.  R2DM_RANGE_SHARED=1 = one slot, as in round 5.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j425; mkdir -p $O
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
cd $R
for i in 1 2 3; do
  for m in sites shared; do
    case $m in sites) E="";; shared) E="R2DM_RANGE_SHARED=1";; esac
    env $E timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench range=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp
for m in sites shared; do
  case $m in sites) E="";; shared) E="R2DM_RANGE_SHARED=1";; esac
  (cd $R && env $E timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -o kt_$m -- python bench.py $A --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$m.json 2> $O/kt_$m.err)
  F=$(find $O -name "kt_${m}_kernel_trace.csv" | head -1)
  python - "$F" $m <<'PY' | tee -a $O/kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "fir_" in n or "proj" in n or "attention" in n:
        g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        d[(n[:44], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(sys.argv[2], k[0], "grid", k[1], "n", len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
  rm -f $F
done
