#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j43; mkdir -p $O
cd $R
for b in 8 2 1; do B=$b timeout 300 python scripts/graph_probe.py 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $O/graph.log
