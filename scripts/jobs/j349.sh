#!/bin/bash
# round 5: neighbour-lane (DPP) variants of out_conv, fir_down2, fir_up2 -- the GPU suite, per-kernel times base | new (rocprofv3, one job), step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j349; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | tail -3
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4 5; do
  R2DM_HIP_LIB=$R/build_probe/lib_base.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line base
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line new
done | tee $O/ab.log
for lib in build_probe/lib_base.so r2dm_amd/libr2dm_hip.so; do
  n=$(basename $lib .so)
  R2DM_HIP_LIB=$R/$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$n -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt_$n.json 2> $O/kt_$n.err
  rm -f $(find $O -name "kt_${n}_kernel_trace.csv")
  f=$(find $O -name "kt_${n}_kernel_stats.csv" | head -1)
  echo "== $n"; python - "$f" <<'PY' | tee $O/kt_$n.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    if 'conv_f16x2' in r['Name']: continue
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
