#!/bin/bash
# GPU job 1 (round 2): counter list, accuracy budget (default / accurate-SiLU / fp32-MFMA algorithm), PMC baseline, bench baseline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j1; mkdir -p $O
cd $R
rocprofv3 -L > $O/counters.txt 2>&1
python scripts/error_budget.py > $O/budget_default.log 2>&1
R2DM_HIP_LIB=$R/build_probe/lib_accsilu.so python scripts/error_budget.py > $O/budget_accsilu.log 2>&1
R2DM_CONV_ALGO=f32 python scripts/error_budget.py > $O/budget_f32.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU -d $O/pmc1 -o pmc1 -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES -d $O/pmc2 -o pmc2 -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/pmc2.log 2>&1
ls -R $O | head -40
