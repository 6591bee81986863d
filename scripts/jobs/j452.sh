#!/bin/bash
# where the waves of the secondary kernels spend their cycles (own PMC pass): SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_LDS, SQ_ACTIVE_INST_VMEM, SQ_ACTIVE_INST_SCA
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j452; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O -o w -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/w.json 2> $O/w.err
tail -2 $O/w.err
python - <<PY
import csv, collections, glob, re
f = glob.glob('$O/**/w_counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = re.sub(r'r2dm::|void |\(.*', '', r['Kernel_Name'])[:44]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:16]:
    wc = max(v.get('SQ_WAVE_CYCLES', 1), 1)
    print('%-44s n %4d ' % (k, n[k]) + ' '.join('%s %.3f' % (c.replace('SQ_', '').replace('ACTIVE_INST_', 'act_'), v[c] / wc) for c in sorted(v) if c != 'SQ_WAVE_CYCLES'))
PY
rm -f $(find $O -name "w_counter_collection.csv")
