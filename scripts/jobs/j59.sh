#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j59; mkdir -p $O
cd $R
for cfg in "1 L4_512_512" "8 L1_64_64"; do set -- $cfg
B=$1 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=2000 SHAPES=$2 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl_$1_$2.log
echo "== B=$1 $2"; sed -n 1,1p $O/tl_$1_$2.log; grep "  M  M epi" $O/tl_$1_$2.log | head -24
done
