#!/bin/bash
# round 5 (third session): the small kernels of a step -- FIR down-sampler leaves the GroupNorm statistics of its output (3 gn_partial + 3 gn_finalize
# launches gone), in_conv through the scalar cache + residual a block ahead + channel shares, out_conv with hand-counted double-buffered loads --
# GPU suite on the new code, then alternating bench lines base | new in one job, then a kernel trace of the new code
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j340; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
for i in 1 2 3; do
  for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$lib python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"
  done
done | tee $O/ab.log
for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so; do
  n=$(basename $lib .so)
  R2DM_HIP_LIB=$R/$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$n -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt_$n.json 2> $O/kt_$n.err
  rm -f $(find $O -name "kt_${n}_kernel_trace.csv")
  f=$(find $O -name "kt_${n}_kernel_stats.csv" | head -1)
  echo "== $n"; python - "$f" <<'PY' | tee $O/kt_$n.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
