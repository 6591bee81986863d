#!/bin/bash
# round 3: would half-batches at level 1 (working set inside the 256 MB Infinity Cache) pay?  Per-launch times at batch 4 vs batch 8.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j93; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --prewarm-s 0.5 --steps 16"
for B in 4 8; do
rocprofv3 --kernel-trace --output-format csv -d $O -o kt_b$B -- python $R/bench.py $A --batch $B > $O/b$B.json 2> $O/b$B.err
python $R/scripts/per_shape_table.py $O/kt_b${B}_kernel_trace.csv $B > $O/shapes_b$B.txt; grep -A20 "^# by shape" $O/shapes_b$B.txt | cut -c1-100
done
rm -f $O/*kernel_trace.csv
