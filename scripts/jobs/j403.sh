#!/bin/bash
# round 6: Infinity-Cache-aware walks of the consumers of >256 MB tensors -- fir_down2_stats against the down-sampling convolution's walk, the 1x1 skip convolution against conv1's.
# alternating A/B in one job (step time), then per-kernel times from a kernel trace of each setting
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j403; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -x -k "fir or unet_golden or north_star or batch8" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in old new firnew projnew; do
    case $m in old) E="R2DM_FIR_ORDER=0 R2DM_PROJ_ORDER=0";; new) E="";; firnew) E="R2DM_PROJ_ORDER=0";; projnew) E="R2DM_FIR_ORDER=0";; esac
    env $E timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp
for m in old new; do
  case $m in old) E="R2DM_FIR_ORDER=0 R2DM_PROJ_ORDER=0";; new) E="";; esac
  env $E timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -o kt_$m -- python $R/bench.py $A --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$m.json 2> $O/kt_$m.err
  F=$(find $O -name "kt_${m}_kernel_trace.csv" | head -1)
  python - "$F" $m <<'PY' | tee -a $O/kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "fir_down2" in n or "proj" in n or "fir_up2" in n:
        g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        d[(n[:40], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(sys.argv[2], k[0], "grid", k[1], "n", len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
  rm -f $F
done
