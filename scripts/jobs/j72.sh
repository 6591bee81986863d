#!/bin/bash
# final-state evidence: GPU suite (default and R2DM_CONV_ALGO=f32), j35-style profile set (r02h: final round-2 code)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j72; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp
python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python $R/bench.py --steps 256 --warmup 8 --no-cpu-baseline --no-torch-baseline > $O/bench_256.json 2> $O/bench_256.err
python $R/bench.py --config 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench_kt -- python $R/bench.py --no-cpu-baseline --no-torch-baseline > $O/bench_kt.json 2> $O/bench_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o bench_fetch -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --steps 4 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o bench_write -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --steps 4 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o bench_mfma -- python $R/bench.py --no-cpu-baseline --no-torch-baseline --steps 4 --warmup 1 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
python - <<PY
import json
for f in ("bench_n1", "bench_256", "bench_c2", "bench_c4"):
    try:
        j = json.load(open("$O/%s.json" % f)); r = j["roofline"]
        print(f, "value", round(j["value"], 3), "ms/step", round(j["ms_per_step"], 3), "frac", round(r["frac"], 3), "torch", (j.get("torch_rocm_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
