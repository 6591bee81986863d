#!/bin/bash
# round 3 evidence set (r03b): smoke, sampler tests, bench lines, rocprofv3 kernel trace + three PMC passes, 256-step / DDIM-32 /
# 128x2048 validation, per-step times
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j84; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_configs.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp
python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-exact-baseline > $O/bench_20.json 2> $O/bench_20.err
python $R/bench.py --steps 256 --warmup 8 --no-cpu-baseline --no-torch-baseline --no-exact-baseline > $O/bench_256.json 2> $O/bench_256.err
python $R/bench.py --config 2 --no-cpu-baseline --no-exact-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-exact-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python $R/bench.py --precision fp16 --no-cpu-baseline --no-torch-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench_kt -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --no-exact-baseline > $O/bench_kt.json 2> $O/bench_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o bench_fetch -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --no-exact-baseline --steps 4 --warmup 1 --prewarm-s 0.5 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o bench_write -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --no-exact-baseline --steps 4 --warmup 1 --prewarm-s 0.5 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o bench_mfma -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --no-exact-baseline --steps 4 --warmup 1 --prewarm-s 0.5 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
cd $R
{ STEPS=256 timeout 300 python scripts/step_times.py; } 2>&1 | grep -v amdgpu.ids | tee $O/step_times.log | cut -c1-300
{ timeout 900 python scripts/validate_256.py; } 2>&1 | grep -v amdgpu.ids | tee $O/validate_256.log
python - <<PY
import json
for f in ("bench_n1", "bench_20", "bench_256", "bench_c2", "bench_c4", "bench_fp16"):
    try:
        j = json.load(open("$O/%s.json" % f)); r = j["roofline"]
        print(f, "value", round(j["value"], 3), "ms/step", round(j["ms_per_step"], 3), "frac", round(r["frac"], 3), "vs", j.get("vs_baseline"), "exact", (j.get("exact_split_baseline") or {}).get("value"), "sclk", r["board"]["sclk_mhz"])
    except Exception as e:
        print(f, "FAILED", e)
PY
