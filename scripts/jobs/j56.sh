#!/bin/bash
# pixel-aligned three-way splits of the stagers' transform (2|1|1 and 1|2|1 pixels): timeline + A/B against the committed kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j56; mkdir -p $O
cd $R
for v in k812 k412; do
R2DM_HIP_LIB=$R/build_probe/lib_${v}_prof.so MAXEV=120 SHAPES=L1_64_64 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/timeline_$v.log
echo "== $v"; sed -n 1,1p $O/timeline_$v.log; sed -n 60,100p $O/timeline_$v.log
done
for rep in 1 2; do for lib in base k812 k412; do
R2DM_HIP_LIB=$R/build_probe/lib_$lib.so timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-torch-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench $lib', round(j['value'],3), round(j['ms_per_step'],3), round(j['roofline']['dominant_kernel']['ms_per_step'],3))"; done; done
