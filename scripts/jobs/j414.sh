#!/bin/bash
# round 6: u_block4's 32-channel-tile launches through the operand pre-pass with the GroupNorm folded into the pass (presplit_fold_kernel) -- bit-identity, A/B, per-shape times
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j414; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_range.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5; do
  for m in 0 1; do
    R2DM_F2_PRESPLIT_NARROW=$m timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench narrow_presplit=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
cd /tmp
for m in 0 1; do
R2DM_F2_PRESPLIT_NARROW=$m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt$m -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt$m.json 2> $O/kt$m.err
python $R/scripts/per_shape_table.py $(find $O -name "kt${m}_kernel_trace.csv" | head -1) > $O/conv_shapes$m.txt 2>&1
rm -f $(find $O -name "kt${m}_kernel_trace.csv")
echo "== narrow_presplit=$m"; grep "8x128" $O/conv_shapes$m.txt | tail -3; grep -i "presplit\|finalize" $(find $O -name "kt${m}_kernel_stats.csv" | head -1) | cut -c1-140
done
