#!/bin/bash
# round 5: Infinity-Cache residency -- the step and every conv launch at batch 4 against batch 8 (is 2 t(4) < 0.95 t(8) anywhere?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j333; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
for i in 1 2 3; do
  for b in 8 4; do
    python $R/bench.py $A --batch $b --prewarm-s 1.0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('batch $b', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s')"
  done
done | tee $O/steps.log
for b in 8 4; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_b$b -- python $R/bench.py $A --batch $b --prewarm-s 0.5 > $O/kt_b$b.json 2> $O/kt_b$b.err
  python $R/scripts/per_shape_table.py $(find $O -name "kt_b${b}_kernel_trace.csv" | head -1) > $O/conv_shapes_b$b.txt 2>&1
  rm -f $(find $O -name "kt_b${b}_kernel_trace.csv")
done
paste <(awk '!/^#/{print $1, $2, $3, $4, $5, $6, $(NF-3)}' $O/conv_shapes_b8.txt) <(awk '!/^#/{print $(NF-3)}' $O/conv_shapes_b4.txt) | awk '{t8=$(NF-1); t4=$NF; printf "%s  t8 %.1f  2*t4 %.1f  ratio %.3f\n", $0, t8, 2*t4, 2*t4/t8}' | tee $O/compare.txt | head -60
