#!/bin/bash
# round 3: attention core without the rescale when the running maximum did not move: identical output? time?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j113; mkdir -p $O
cd /tmp
for lib in build_probe/lib_dc_u4.so r2dm_amd/libr2dm_hip.so; do
n=$(basename $lib .so); rm -rf /tmp/prof_$n
R2DM_HIP_LIB=$R/$lib timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0.5 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$n.json 2>/dev/null
f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
echo "$n: $(python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attention' in r['Name']: print(r['Name'][11:40], round(float(r['AverageNs'])/1e3,1), 'us;', end=' ')
") ms/step $(python -c "import json; print(round(json.load(open('/tmp/b_$n.json'))['ms_per_step'],3))")"
done 2>&1 | tee $O/variants.log
