#!/bin/bash
# round 4: robustness on the round's code -- geometry fuzz (200 cases), sampler fuzz, soak (bitwise repeatability), the re-based U-Net bar
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j214; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_unet.py -q -s -k "operand_split_modes" 2>&1 | grep -E "U-Net 64x1024|passed|failed|assert" | tee $O/unet_modes.log
for seed in 2 3 4 5 6; do SEED=$seed CASES=40 timeout 900 python scripts/fuzz_configs.py > $O/fuzz_$seed.log 2>&1; echo "seed $seed: $(grep -c ' OK$' $O/fuzz_$seed.log) ok, $(grep -c FAIL $O/fuzz_$seed.log) FAIL, $(grep -c rejected $O/fuzz_$seed.log) rejected; $(tail -1 $O/fuzz_$seed.log | cut -c1-150)"; grep -E "FAIL|fault|Error" $O/fuzz_$seed.log | cut -c1-250 | head -5; done | tee $O/fuzz_summary.log
CASES=40 timeout 1200 python scripts/fuzz_sampler.py > $O/fuzz_sampler.log 2>&1; tail -4 $O/fuzz_sampler.log | cut -c1-300
timeout 1200 python scripts/soak.py 2>&1 | grep -v amdgpu | tee $O/soak.log
