#!/bin/bash
# LDS bank conflicts of the conv_f16x2 instances (own PMC pass)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j450; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
rocprofv3 --list-avail 2>/dev/null | grep -i "Counter_Name.*LDS" | head -30 > $O/avail.txt
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O -o lds -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/lds.json 2> $O/lds.err
python - <<PY
import csv, collections, glob, re
f = glob.glob('$O/**/lds_counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = re.sub(r'r2dm::|void |\(.*', '', r['Kernel_Name'])[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_INSTS_LDS': n[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:12]:
    print('%-50s n %5d ' % (k, n[k]) + '  '.join('%s %.4g' % (c, v[c] / max(n[k], 1)) for c in sorted(v)) + '  conflict/active %.3f' % (v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
rm -f $(find $O -name "lds_counter_collection.csv")
