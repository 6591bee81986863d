#!/bin/bash
# round 5: conv_f16x2 (6648 packed-fp32 instructions from vector-typed epilogue / transform arithmetic) and proj_f16x2 compiled WITHOUT packed-fp32 instruction selection -- a speed experiment
# (attention got 4 % faster that way): bit-identity of whole forwards (scripts/ab_bits.py), then alternating bench lines default | conv_f16x2 | conv_f16x2 + proj_f16x2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j376; mkdir -p $O; cd $R
rm -f /tmp/ab_bits.pt
OUT=/tmp/ab_bits.pt python scripts/ab_bits.py 2>&1 | grep ab_bits | tee $O/bits.log
R2DM_HIP_LIB=$R/build_probe/lib_nopkcp.so OUT=/tmp/ab_bits.pt python scripts/ab_bits.py 2>&1 | grep ab_bits | tee -a $O/bits.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4; do
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line default
  R2DM_HIP_LIB=$R/build_probe/lib_nopkc.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line nopk_conv
  R2DM_HIP_LIB=$R/build_probe/lib_nopkcp.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line nopk_conv_proj
done | tee $O/ab.log
