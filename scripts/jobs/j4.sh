#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j4; mkdir -p $O
cd $R/scripts
R2DM_HIP_LIB=$R/build_probe/lib_duo_prof.so timeout 120 python duo_timeline.py > $O/timeline.log 2>&1
head -150 $O/timeline.log
