#!/bin/bash
# round 3, final code: geometry fuzz (new seeds) + sampler fuzz + soak
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j119; mkdir -p $O
cd $R
SEED=7 CASES=120 timeout 1500 python scripts/fuzz_configs.py 2>&1 | grep -v amdgpu.ids > $O/fuzz_configs_seed7.log; tail -1 $O/fuzz_configs_seed7.log; grep -c "FAIL" $O/fuzz_configs_seed7.log
SEED=11 CASES=40 timeout 900 python scripts/fuzz_sampler.py 2>&1 | grep -v amdgpu.ids > $O/fuzz_sampler_seed11.log; tail -1 $O/fuzz_sampler_seed11.log; grep "FAIL" $O/fuzz_sampler_seed11.log | head
timeout 900 python scripts/soak.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.log | tail -6
