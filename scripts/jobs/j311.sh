#!/bin/bash
# round 5: where the fp16 bulk mode spends its step (rocprofv3 kernel trace + per-launch table)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j311; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench_kt -- python $R/bench.py --precision fp16 --no-cpu-baseline --no-torch-baseline --steps 24 --warmup 2 --prewarm-s 0.5 > $O/bench_kt.json 2> $O/bench_kt.err
python $R/scripts/per_shape_table.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/shapes_fp16.txt 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fp16.csv
rm -rf $O/kt
head -25 $O/kernel_stats_fp16.csv | cut -c1-120
