#!/bin/bash
# out_conv with the neighbour pixels by lane exchange (R2DM_OUTCONV_SHFL=1): bit-identity of a forward, kernel time by rocprofv3, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j453; mkdir -p $O
cd $R
rm -f /tmp/ab.pt
OUT=/tmp/ab.pt R2DM_OUTCONV_SHFL=0 timeout 300 python scripts/ab_bits.py 2>&1 | grep ab_bits | tee $O/bits.log
OUT=/tmp/ab.pt R2DM_OUTCONV_SHFL=1 timeout 300 python scripts/ab_bits.py 2>&1 | grep ab_bits | tee -a $O/bits.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for m in 0 1; do
    R2DM_OUTCONV_SHFL=$m timeout 300 python bench.py $A --steps 128 --warmup 4 2>$O/err_$m.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench outconv_shfl=$m', round(j['ms_per_step'],3), round(j['value'],3))"
  done
done | tee $O/ab.log
cd /tmp
for m in 0 1; do
  R2DM_OUTCONV_SHFL=$m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$m -- python $R/bench.py $A --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$m.json 2> $O/kt_$m.err
  rm -f $(find $O -name "kt_${m}_kernel_trace.csv")
  echo "shfl=$m $(grep 'conv_direct_rows' $(find $O -name "kt_${m}_kernel_stats.csv") | cut -d, -f1-4 | cut -c1-100)"
done | tee $O/oc.log
