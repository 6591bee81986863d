#!/bin/bash
# proj_f16x2: channel-slot rotation in the x tile (no 4-way write conflicts): parity, kernel time old | new by rocprofv3, conflicts by PMC, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j451; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_fp16_mode.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for l in fin sw; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp
for l in fin sw; do
  R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$l -- python $R/bench.py $A --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$l.json 2> $O/kt_$l.err
  rm -f $(find $O -name "kt_${l}_kernel_trace.csv")
  echo "$l $(grep 'proj_f16x2_kernel' $(find $O -name "kt_${l}_kernel_stats.csv") | cut -d, -f1-4 | cut -c1-90)"
done | tee $O/proj.log
R2DM_HIP_LIB=$R/build_probe/lib_sw.so timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O -o lds -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/lds.json 2> $O/lds.err
python - <<PY | tee -a $O/proj.log
import csv, collections, glob, re
f = glob.glob('$O/**/lds_counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = re.sub(r'r2dm::|void |\(.*', '', r['Kernel_Name'])[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'proj' in k: print(k, 'conflict/active %.3f' % (v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1)))
PY
rm -f $(find $O -name "lds_counter_collection.csv")
