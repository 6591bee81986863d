#!/bin/bash
# round 3: in_conv (conv_few_in_kernel) with its output channels split over gridDim.y: 1 / 2 / 4 / 8 ways
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j110; mkdir -p $O
cd /tmp
for ny in 1 2 4 8; do
rm -rf /tmp/prof_$ny
R2DM_FEW_IN_SPLIT=$ny timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$ny -o p -- python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0.5 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$ny.json 2>/dev/null
f=$(find /tmp/prof_$ny -name "*kernel_stats.csv" | head -1)
echo "split $ny: $(grep -E 'conv_few_in|conv_direct_rows' $f | awk -F, '{gsub(/"/,""); print $1, "calls", $2, "avg_ns", $4}' | tr '\n' ';') ms/step $(python -c "import json; print(round(json.load(open('/tmp/b_$ny.json'))['ms_per_step'],3))")"
done 2>&1 | tee $O/split.log
