#!/bin/bash
# round 3 evidence set r03f (final round-3 code: after the attention and 1x1 work): rocprofv3 kernel trace + three PMC passes + bench lines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j125; mkdir -p $O
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench_kt -- python $R/bench.py $A --prewarm-s 0.5 > $O/bench_kt.json 2> $O/bench_kt.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o bench_fetch -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o bench_write -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/pmc_write.json 2> $O/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o bench_mfma -- python $R/bench.py $A --steps 4 --warmup 1 --prewarm-s 0.1 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
timeout 600 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python $R/bench.py --steps 256 --warmup 8 $A > $O/bench_256.json 2> $O/bench_256.err
timeout 600 python $R/bench.py --config 2 --no-cpu-baseline --no-exact-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-exact-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python $R/bench.py --precision fp16 --no-cpu-baseline --no-torch-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
timeout 600 python $R/bench.py --steps 20 --warmup 2 $A > $O/bench_20.json 2> $O/bench_20.err
cd $R; timeout 600 python -m pytest tests/test_hip_configs.py -q -m gpu -k "rccl" 2>&1 | tail -2
du -sh $O; ls -la $O | awk '{print $5, $9}' | sort -n | tail -6
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
