#!/bin/bash
# round 5, last session: robustness re-run on the round's FINAL code (after the packed-fp32 build change touched attention / in_conv / out_conv / FIR / posterior / norm / embed):
# geometry fuzz on NEW seeds (3 x 40 cases vs the fp64 oracle) and the sampler fuzz (new seed)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j378; mkdir -p $O
cd $R
for seed in 31 32 33; do SEED=$seed CASES=40 timeout 500 python scripts/fuzz_configs.py > $O/fuzz_$seed.log 2>&1; echo "seed $seed: $(grep -c ' OK$' $O/fuzz_$seed.log) ok, $(grep -c FAIL $O/fuzz_$seed.log) FAIL, $(grep -c rejected $O/fuzz_$seed.log) rejected; $(tail -1 $O/fuzz_$seed.log | cut -c1-200)" | tee -a $O/fuzz_summary.log; done
SEED=7 CASES=40 timeout 500 python scripts/fuzz_sampler.py > $O/fuzz_sampler.log 2>&1; tail -4 $O/fuzz_sampler.log | cut -c1-300 | tee -a $O/fuzz_summary.log
