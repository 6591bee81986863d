#!/bin/bash
# serialised loads found in the ISA (attention: the Q operand, 32 round trips before the first key tile; FIR resamplers: one exec-masked block + full wait per input row) now batched:
# parity of the kernels, A/B on the step, per-kernel times by rocprofv3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j441; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_fp16_mode.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for l in head ld; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp
for l in head ld; do
  R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$l -- python $R/bench.py $A --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$l.json 2> $O/kt_$l.err
  rm -f $(find $O -name "kt_${l}_kernel_trace.csv")
  echo "== $l"; python - <<PY
import csv,glob,re
f=glob.glob('$O/**/kt_${l}_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=re.sub(r'r2dm::|void ','',r['Name'])
    if re.search(r'attention|fir_', n): print('%-70s calls %5s avg %8.1f us' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done | tee $O/kernels.log
