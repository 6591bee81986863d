#!/bin/bash
# round 3: attention core, timing ablations (wrong results): what bounds the loop?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j122; mkdir -p $O
cd /tmp
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_att_nostage.so build_probe/lib_att_nosoftmax.so build_probe/lib_att_nobarrier.so build_probe/lib_att_nostage_nosoftmax.so; do
n=$(basename $lib .so); rm -rf /tmp/prof_$n
R2DM_HIP_LIB=$R/$lib timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $R/bench.py --steps 4 --warmup 1 --prewarm-s 0.3 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$n.json 2>/dev/null
f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
echo "$n: $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'attention' in r['Name']: print(r['Name'][11:40], round(float(r['AverageNs'])/1e3,1), 'us;', end=' ')
")"
done 2>&1 | tee $O/ablation.log
