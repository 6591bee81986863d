#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j100; mkdir -p $O
cd $R; R2DM_DEBUG_SYNC=1 SEED=2 CASES=24 timeout 1500 python scripts/fuzz_configs.py > $O/fuzz_dbg.log 2>&1; grep -v amdgpu.ids $O/fuzz_dbg.log | tail -25 | cut -c1-220; grep -c "^case" $O/fuzz_dbg.log
