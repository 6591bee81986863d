#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j212; mkdir -p $O
cd $R
timeout 600 python scripts/two_stream_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/two_stream.log
