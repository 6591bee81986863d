#!/bin/bash
# kernel-trace statistics of one kernel name pattern ($PAT) for the libraries in $LIBS, plus the GroupNorm / unet tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j67; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp
for lib in $LIBS; do
n=$(basename $lib .so)
R2DM_HIP_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$n -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --steps 32 > $O/kt_$n.json 2> $O/kt_$n.err
echo "== $lib"; grep -E "$PAT" $O/kt_${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
python -c "import json; j=json.load(open('$O/kt_$n.json')); print('ms/step', round(j['ms_per_step'],3))"
done
