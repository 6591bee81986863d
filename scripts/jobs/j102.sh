#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j102; mkdir -p $O
cd $R; SEED=1 CASES=40 timeout 1500 python scripts/fuzz_sampler.py > $O/fuzz_sampler.log 2>&1; grep -v amdgpu.ids $O/fuzz_sampler.log | grep -E "FAIL|cases,|Error|Traceback|bad arguments" | cut -c1-250 | head -20; grep -v amdgpu.ids $O/fuzz_sampler.log | tail -4 | cut -c1-250
