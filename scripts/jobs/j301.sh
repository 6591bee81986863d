#!/bin/bash
# round 5: per-launch tables (rocprofv3 kernel trace of bench.py) with the 64 x 8 tile off / on in one job; range-guard replay tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j301; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for m in 0 1; do
  R2DM_F2_TALL=$m timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt$m -o bench_kt -- python $R/bench.py $A --steps 24 --warmup 2 --prewarm-s 0.5 > $O/bench_kt$m.json 2> $O/bench_kt$m.err
  f=$(find $O/kt$m -name "*kernel_trace.csv" | head -1)
  python $R/scripts/per_shape_table.py $f > $O/shapes_tall$m.txt 2>&1
  rm -rf $O/kt$m
done
cd $R
timeout 900 python -m pytest tests/test_hip_range.py -q -x > $O/pytest_range.log 2>&1; tail -5 $O/pytest_range.log
