#!/bin/bash
# kernel times of the fp16 bulk mode (rocprofv3 --kernel-trace --stats of bench.py --precision fp16)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j444; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt16 -- python $R/bench.py $A --precision fp16 --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt16.json 2> $O/kt16.err
rm -f $(find $O -name "kt16_kernel_trace.csv")
python - <<PY
import csv,glob,re
f=glob.glob('$O/**/kt16_kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=[int(r['Calls']) for r in rows if 'posterior' in r['Name']][0]
for r in rows[:26]:
    n=re.sub(r'r2dm::|void ','',r['Name'])
    print('%-72s per step %5.1f avg %8.1f us  total/step %7.1f us' % (n[:72], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/steps))
PY
