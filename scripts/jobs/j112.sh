#!/bin/bash
# round 3: in_conv with its residual loads hoisted; out_conv with deeper unrolling / one row per thread (rocprofv3 averages)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j112; mkdir -p $O
cd /tmp
for lib in build_probe/lib_dc_u4.so r2dm_amd/libr2dm_hip.so; do
n=$(basename $lib .so); rm -rf /tmp/prof_$n
R2DM_HIP_LIB=$R/$lib timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0.5 --no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs > /tmp/b_$n.json 2>/dev/null
f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
echo "$n: $(grep -E 'conv_few_in|conv_direct_rows' $f | awk -F, '{gsub(/"/,""); print $1, "avg_ns", $4}' | sed 's/void r2dm:://; s/(r2dm::ConvParams)//' | tr '\n' ';') ms/step $(python -c "import json; print(round(json.load(open('/tmp/b_$n.json'))['ms_per_step'],3))")"
done 2>&1 | tee $O/variants.log
cd $R; timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -m gpu -x -q 2>&1 | tail -2
