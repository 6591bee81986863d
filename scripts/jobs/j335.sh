#!/bin/bash
# round 5: the staging waves' transform on pixel pairs (packed fp32 instructions) against the scalar form (-DF2_XF_SCALAR library):
# bit-identity of whole forwards, then alternating bench lines in the parity mode and in the fp16 bulk mode
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j335; mkdir -p $O; cd $R
rm -f /tmp/ab_bits.pt
R2DM_HIP_LIB=$R/build_probe/lib_xfscalar.so OUT=/tmp/ab_bits.pt python scripts/ab_bits.py 2>&1 | grep ab_bits | tee $O/bits.log
OUT=/tmp/ab_bits.pt python scripts/ab_bits.py 2>&1 | grep ab_bits | tee -a $O/bits.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
for prec in fp32 fp16; do
for i in 1 2 3 4; do
  for lib in build_probe/lib_xfscalar.so r2dm_amd/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$lib python $R/bench.py $A --precision $prec --prewarm-s 1.0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$prec $lib', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"
  done
done
done | tee $O/ab.log
