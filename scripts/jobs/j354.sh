#!/bin/bash
# round 5, final code: robustness -- geometry fuzz (5 seeds x 40 cases: every kernel-family / tile / statistics-path switch point incl. the round's
# new ones: 64 x 8 and 32 x 4 tiles, folded GroupNorm, the down-sampler's statistics variant, in_conv's channel shares), sampler fuzz, soak
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j354; mkdir -p $O
cd $R
for seed in 21 22 23 24 25; do SEED=$seed CASES=40 timeout 900 python scripts/fuzz_configs.py > $O/fuzz_$seed.log 2>&1; echo "seed $seed: $(grep -c ' OK$' $O/fuzz_$seed.log) ok, $(grep -c FAIL $O/fuzz_$seed.log) FAIL, $(grep -c rejected $O/fuzz_$seed.log) rejected; $(tail -1 $O/fuzz_$seed.log | cut -c1-150)"; grep -E "FAIL|fault|Error" $O/fuzz_$seed.log | cut -c1-250 | head -5; done | tee $O/fuzz_summary.log
CASES=40 timeout 1200 python scripts/fuzz_sampler.py > $O/fuzz_sampler.log 2>&1; tail -4 $O/fuzz_sampler.log | cut -c1-300 | tee -a $O/fuzz_summary.log
timeout 1200 python scripts/soak.py 2>&1 | grep -v amdgpu | tee $O/soak.log
