#!/bin/bash
# round 6: the shared tile end (EPI3: the staging partner finishes the last 4 (64 x 8 tile) / 2 (128 x 4 tile) quarters of a tile) -- correctness first, then A/B against -DF2_NO_EPI3, per-shape times, timelines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j408; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3; do
  for l in build_probe/lib_noepi3.so r2dm_amd/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$l timeout 300 python bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
for s in L1_64_64 L1_64_64_nores L2_128_128; do
  B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=900 SHAPES=$s timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$s.log
  head -1 $O/tl_$s.log; grep "epi begin\|epi end" $O/tl_$s.log | awk '{print $1}' | paste - - | awk '{print "tile end", $2 - $1}'
done | tee $O/tl_summary.txt
cd /tmp
for l in build_probe/lib_noepi3.so r2dm_amd/libr2dm_hip.so; do
  n=$(basename $l .so)
  R2DM_HIP_LIB=$R/$l timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$n -- python $R/bench.py $A --prewarm-s 0.5 > $O/kt_$n.json 2> $O/kt_$n.err
  python $R/scripts/per_shape_table.py $(find $O -name "kt_${n}_kernel_trace.csv" | head -1) > $O/conv_shapes_$n.txt 2>&1
  rm -f $(find $O -name "kt_${n}_kernel_trace.csv")
  tail -25 $O/conv_shapes_$n.txt | head -22
done
