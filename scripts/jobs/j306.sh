#!/bin/bash
# round 5 evidence set (r05a: before the 32-channel tile; r05b: final code): rocprofv3 kernel trace + three PMC passes + bench lines + whole GPU suite + 256-step validation, on the round's code
# (64 x 8 tile, round-5 tile end of the one-accumulator tiles, GroupNorm folded into its consumer)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${JOB:-j306}; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench_kt -- python $R/bench.py $A --prewarm-s 0.5 > $O/bench_kt.json 2> $O/bench_kt.err
python $R/scripts/per_shape_table.py $(find $O -name "bench_kt_kernel_trace.csv" | head -1) > $O/conv_shapes.txt 2>&1
rm -f $(find $O -name "bench_kt_kernel_trace.csv")
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o bench_fetch -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o bench_write -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/pmc_write.json 2> $O/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o bench_mfma -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python $R/bench.py --steps 256 --warmup 8 $A > $O/bench_256.json 2> $O/bench_256.err
timeout 600 python $R/bench.py --config 2 --no-cpu-baseline --no-exact-baseline --no-compile-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-exact-baseline --no-compile-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python $R/bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
du -sh $O; ls -la $O | awk '{print $5, $9}' | sort -n | tail -4
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2700 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | tee $O/validate_256.log
MODE=ddim STEPS=32 timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/validate_256.log
