#!/bin/bash
# fp16-input out_conv with batched raw loads (a loop of its own): exactness tests, fp16-mode A/B, kernel time
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j445; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_fp16_mode.py tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for l in head2 oc16; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --precision fp16 --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
for l in head2 oc16; do R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp32 lib=$l', round(j['ms_per_step'],3), round(j['value'],3))"; done | tee -a $O/ab.log
cd /tmp
R2DM_HIP_LIB=$R/build_probe/lib_oc16.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt16 -- python $R/bench.py $A --precision fp16 --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt16.json 2> $O/kt16.err
rm -f $(find $O -name "kt16_kernel_trace.csv")
grep "conv_direct_rows\|conv_few_in" $(find $O -name "kt16_kernel_stats.csv") | cut -c1-120 | tee $O/oc.log
