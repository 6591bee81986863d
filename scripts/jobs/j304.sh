#!/bin/bash
# round 5: GroupNorm folded into its consumer -- bit identity, whole-net / range / kernel tests, step A/B against R2DM_GN_FOLD=0, launch counts
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j304; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_unet.py -q -x -k "folded" > $O/pytest_fold.log 2>&1; tail -4 $O/pytest_fold.log
timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_range.py tests/test_hip_kernels.py -q -x -k "golden or range or group_norm or full_size or fold" > $O/pytest_more.log 2>&1; tail -4 $O/pytest_more.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in 0 1; do
    R2DM_GN_FOLD=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fold=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_fold.log
cd /tmp
for m in 0 1; do
R2DM_GN_FOLD=$m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$m -o bench_kt -- python $R/bench.py $A --steps 24 --warmup 2 --prewarm-s 0.5 > $O/bench_kt$m.json 2> $O/bench_kt$m.err
f=$(find $O/kt$m -name "*kernel_trace.csv" | head -1)
python $R/scripts/per_shape_table.py $f > $O/shapes_fold$m.txt 2>&1
cp $(find $O/kt$m -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fold$m.csv
rm -rf $O/kt$m
grep "all 54" $O/shapes_fold$m.txt; grep -i "gn_" $O/kernel_stats_fold$m.csv | cut -c1-150
done
