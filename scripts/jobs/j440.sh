#!/bin/bash
# instruction cache of the conv_f16x2 instances (20 k / 10.7 k / 8.6 k instructions): SQC_ICACHE_REQ / HITS / MISSES per kernel, own PMC pass
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j440; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|SQ_INSTS_SALU\|SQ_INST_CYCLES\|SQ_WAIT_INST\|IFETCH" | head -20 > $O/avail.txt
timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $O -o ic -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/ic.json 2> $O/ic.err
python - <<PY
import csv, collections, glob, re
f = glob.glob('$O/**/ic_counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = re.sub(r'r2dm::|void |\(.*', '', r['Kernel_Name'])[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    if r['Counter_Name'] == 'SQC_ICACHE_REQ': n[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQC_ICACHE_REQ', 0))[:14]:
    q = v.get('SQC_ICACHE_REQ', 0); print('%-62s n %5d  req/launch %10.0f  hit %.4f  miss/launch %9.0f' % (k, n[k], q / max(n[k], 1), v.get('SQC_ICACHE_HITS', 0) / max(q, 1), v.get('SQC_ICACHE_MISSES', 0) / max(n[k], 1)))
PY
rm -f $(find $O -name "ic_counter_collection.csv")
