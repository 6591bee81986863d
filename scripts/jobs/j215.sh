#!/bin/bash
# round 4: geometry fuzz + sampler fuzz with the 128-channel tile FORCED wherever the shape allows (any tile count, any Cin): stress of the MRK = 4 kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j215; mkdir -p $O
cd $R
export R2DM_F2_CO_TILE=128
for seed in 12 13 14; do SEED=$seed CASES=40 timeout 900 python scripts/fuzz_configs.py > $O/fuzz_$seed.log 2>&1; echo "seed $seed: $(grep -c ' OK$' $O/fuzz_$seed.log) ok, $(grep -c FAIL $O/fuzz_$seed.log) FAIL, $(grep -c rejected $O/fuzz_$seed.log) rejected; $(tail -1 $O/fuzz_$seed.log | cut -c1-150)"; grep -E "FAIL|fault|Error" $O/fuzz_$seed.log | cut -c1-250 | head -5; done | tee $O/fuzz_summary.log
CASES=20 timeout 1200 python scripts/fuzz_sampler.py > $O/fuzz_sampler.log 2>&1; tail -2 $O/fuzz_sampler.log | cut -c1-300
timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_configs.py -q -x -k "not second_process and not two_rank and not rccl" 2>&1 | tail -3 | tee $O/pytest_forced_wide.log
