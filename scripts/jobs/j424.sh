#!/bin/bash
# round 6: like for like on ONE box -- the round-5 tree (fca6c3b, built under build_probe/r05tree) against the round-6 tree: alternating bench lines and a kernel trace of each (per-kernel averages)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j424; mkdir -p $O
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for t in build_probe/r05tree .; do
    (cd $R/$t && timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tree=$t', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))")
  done
done | tee $O/ab.log
cd /tmp
for t in build_probe/r05tree .; do
  n=$(echo $t | tr -c 'a-z0-9' '_')
  (cd $R/$t && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$n -- python bench.py $A --prewarm-s 0.5 > $O/kt_$n.json 2> $O/kt_$n.err)
  rm -f $(find $O -name "kt_${n}_kernel_trace.csv")
  echo "== tree $t"; python - $(find $O -name "kt_${n}_kernel_stats.csv" | head -1) <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ns = sum(int(r['Calls']) for r in rows if 'posterior_kernel' in r['Name'])
for r in rows[:24]:
    name = re.sub(r'r2dm::|void |\(anonymous namespace\)::', '', r['Name'])[:70]
    print("%-70s %5.1f/step %8.1f us avg %8.1f us/step" % (name, int(r['Calls']) / ns, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / ns))
print("all kernels us/step", sum(float(r['TotalDurationNs']) for r in rows) / 1e3 / ns)
PY
done | tee $O/kernels.txt
