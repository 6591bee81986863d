#!/bin/bash
# round 5 (third session): small-kernel changes, alternating bench lines base (HEAD e7bcddb) | new in one job, the in_conv channel shares, kernel trace of the new code
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j343; mkdir -p $O; cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4; do
  R2DM_HIP_LIB=$R/build_probe/lib_base.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line base
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line new
  R2DM_FEW_IN_SPLIT=4 python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line new_split4
  R2DM_FIR_STATS=0 python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line new_firstats0
done | tee $O/ab.log
for v in 2 4 1; do
  R2DM_FEW_IN_SPLIT=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$v -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt_$v.json 2> $O/kt_$v.err
  rm -f $(find $O -name "kt_${v}_kernel_trace.csv")
  f=$(find $O -name "kt_${v}_kernel_stats.csv" | head -1)
  echo "== in_conv shares $v"; python - "$f" <<'PY' | tee $O/kt_$v.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
