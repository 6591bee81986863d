#!/bin/bash
# round 5: the tile end of the one-accumulator tiles (LDS bias table, DMA-prefetched residual, no waiting row block) -- tests, step A/B against -DF2_EPI_V1, timeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j303; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "conv" > $O/pytest_conv.log 2>&1; tail -4 $O/pytest_conv.log
timeout 900 python -m pytest tests/test_hip_unet.py -q -x > $O/pytest_unet.log 2>&1; tail -3 $O/pytest_unet.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for lib in build_probe/lib_epi_v1.so r2dm_amd/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$lib timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $lib', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_epi.log
for s in L1_64_64 L2_128_128; do
  B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=700 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$s.log
  head -1 $O/tl_$s.log
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o bench_kt -- python $R/bench.py $A --steps 24 --warmup 2 --prewarm-s 0.5 > $O/bench_kt.json 2> $O/bench_kt.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python $R/scripts/per_shape_table.py $f > $O/shapes.txt 2>&1
rm -rf $O/kt
grep "all 54" $O/shapes.txt
