#!/bin/bash
# round 5: 32-channel tiles (Geo<1, 2>) for u_block4 (128 64-channel tiles on 256 CUs) -- kernel tests, whole-net tests, step A/B against R2DM_F2_NARROW=0, per-launch table
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j308; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "conv3x3" > $O/pytest_conv.log 2>&1; tail -4 $O/pytest_conv.log
timeout 900 python -m pytest tests/test_hip_unet.py -q -x > $O/pytest_unet.log 2>&1; tail -3 $O/pytest_unet.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for m in 0 1; do
    R2DM_F2_NARROW=$m timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench narrow=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_narrow.log
cd /tmp
for m in 0 1; do
R2DM_F2_NARROW=$m timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt$m -o bench_kt -- python $R/bench.py $A --steps 24 --warmup 2 --prewarm-s 0.5 > $O/bench_kt$m.json 2> $O/bench_kt$m.err
python $R/scripts/per_shape_table.py $(find $O/kt$m -name "*kernel_trace.csv" | head -1) > $O/shapes_narrow$m.txt 2>&1
rm -rf $O/kt$m
grep "all 54" $O/shapes_narrow$m.txt; sed -n 30,35p $O/shapes_narrow$m.txt
done
