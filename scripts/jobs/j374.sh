#!/bin/bash
# round 5: the CU-sharing kernels with packed-fp32 instructions that take SGPR operands (attention, in_conv / out_conv, FIR, posterior) compiled WITHOUT packed-fp32 instruction selection
# (-Xclang -target-feature -Xclang -packed-fp32-ops: lib_nopk) -- what does it cost?  alternating bench lines, then the per-kernel averages
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j374; mkdir -p $O; cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4; do
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line default
  R2DM_HIP_LIB=$R/build_probe/lib_nopk.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line nopk
done | tee $O/ab.log
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_nopk.so; do
  n=$(basename $lib .so)
  R2DM_HIP_LIB=$R/$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$n -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt_$n.json 2> $O/kt_$n.err
  rm -f $(find $O -name "kt_${n}_kernel_trace.csv")
  f=$(find $O -name "kt_${n}_kernel_stats.csv" | head -1)
  echo "== $n"; python - "$f" <<'PY' | tee $O/kt_$n.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    if any(k in r['Name'] for k in ('attention','conv_direct','conv_few','fir_','posterior')):
        print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
