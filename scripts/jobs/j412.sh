#!/bin/bash
# round 6 experiment: u_block4's 32-channel-tile launches through the operand pre-pass (R2DM_F2_PRESPLIT_NARROW=1: finalize + presplit launches back, stagers DMA-only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j412; mkdir -p $O
cd $R
R2DM_F2_PRESPLIT_NARROW=1 timeout 600 python -m pytest tests/test_hip_unet.py -q -m gpu -x -k "golden or full_size or batch8" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in 0 1; do
    R2DM_F2_PRESPLIT_NARROW=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench narrow_presplit=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp
R2DM_F2_PRESPLIT_NARROW=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt.json 2> $O/kt.err
python $R/scripts/per_shape_table.py $(find $O -name "kt_kernel_trace.csv" | head -1) > $O/conv_shapes.txt 2>&1
rm -f $(find $O -name "kt_kernel_trace.csv")
grep "8x128\|presplit" $O/conv_shapes.txt | head; grep -i "presplit\|finalize" $(find $O -name "kt_kernel_stats.csv" | head -1) | cut -c1-160
