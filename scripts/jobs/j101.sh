#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j101; mkdir -p $O
cd $R
for seed in 2 3 4 5 6; do SEED=$seed CASES=40 timeout 900 python scripts/fuzz_configs.py > $O/fuzz_$seed.log 2>&1; echo "seed $seed: $(grep -c ' OK$' $O/fuzz_$seed.log) ok, $(grep -c FAIL $O/fuzz_$seed.log) FAIL, $(grep -c rejected $O/fuzz_$seed.log) rejected; $(tail -1 $O/fuzz_$seed.log | cut -c1-150)"; grep -E "FAIL|fault|Error" $O/fuzz_$seed.log | cut -c1-250 | head -5; done
