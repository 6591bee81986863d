#!/bin/bash
# in_conv: the eight biases of a block requested early (one wait) -- kernel time by rocprofv3, fp32 mode
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j448; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for l in fin fib fin fib; do
  R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$l -- python $R/bench.py $A --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$l.json 2> $O/kt_$l.err
  rm -f $(find $O -name "kt_${l}_kernel_trace.csv")
  echo "$l $(grep 'conv_few_in' $(find $O -name "kt_${l}_kernel_stats.csv") | cut -d, -f2-4)"
done | tee $O/fi.log
