#!/bin/bash
# fp16-input out_conv: channels per load batch 2 | 4 | 8 -- kernel time by rocprofv3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j446; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for l in oc1 oc2; do
  R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$l -- python $R/bench.py $A --precision fp16 --steps 16 --warmup 2 --prewarm-s 0.5 > $O/kt_$l.json 2> $O/kt_$l.err
  rm -f $(find $O -name "kt_${l}_kernel_trace.csv")
  echo "$l $(grep 'conv_direct_rows' $(find $O -name "kt_${l}_kernel_stats.csv") | cut -d, -f2-4) ms/step $(python -c "import json;print(round(json.loads(open('$O/kt_$l.json').read().strip().splitlines()[-1])['ms_per_step'],3))")"
done | tee $O/oc.log
