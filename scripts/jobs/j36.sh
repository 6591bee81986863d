#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j36; mkdir -p $O
cd $R
for v in f2_nomfma f2_noxf; do
R2DM_HIP_LIB=$R/build_probe/lib_$v.so MAXEV=64 SHAPES=L1_64_64 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/timeline_$v.log
echo "=== $v"; sed -n 1,2p $O/timeline_$v.log; sed -n 30,64p $O/timeline_$v.log
done
