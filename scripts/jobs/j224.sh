#!/bin/bash
# round 4 evidence set r04b: rocprofv3 kernel trace + three PMC passes + bench lines + GPU suite + validation, on the round's code
# (128-channel tiles for Cin <= 256, operand pre-pass off)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j224; mkdir -p $O
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
for i in 1 2 3; do
  for m in 64 auto; do
    if [ $m = 64 ]; then export R2DM_F2_CO_TILE=64; else unset R2DM_F2_CO_TILE; fi
    timeout 300 python $R/bench.py $A --steps 64 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench tile=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_wide.log
unset R2DM_F2_CO_TILE
du -sh $O; ls -la $O | awk '{print $5, $9}' | sort -n | tail -4
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2700 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for t in auto 128; do
  if [ $t = 128 ]; then export R2DM_F2_CO_TILE=128; else unset R2DM_F2_CO_TILE; fi
  timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/co_tile=$t: /"
done | tee $O/validate_256.log
unset R2DM_F2_CO_TILE
MODE=ddim STEPS=32 timeout 900 python scripts/validate_256.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/validate_256.log
