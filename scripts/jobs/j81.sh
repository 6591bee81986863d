#!/bin/bash
# round 3 checkpoint: the whole GPU suite on the round-3 library, then the bench lines (default with both baselines, 256 steps,
# configs 2 and 4, the fp16 bulk mode)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j81; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x -s > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "differing next|fp16 mode|48-step|gains" $O/pytest.log | cut -c1-200
cd /tmp
python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-exact-baseline > $O/bench_20.json 2> $O/bench_20.err
python $R/bench.py --steps 256 --warmup 8 --no-cpu-baseline --no-torch-baseline --no-exact-baseline > $O/bench_256.json 2> $O/bench_256.err
python $R/bench.py --config 2 --no-cpu-baseline --no-exact-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-exact-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python $R/bench.py --precision fp16 --no-cpu-baseline --no-torch-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
python - <<PY
import json
for f in ("bench_n1", "bench_20", "bench_256", "bench_c2", "bench_c4", "bench_fp16"):
    try:
        j = json.load(open("$O/%s.json" % f)); r = j["roofline"]
        print(f, "value", round(j["value"], 3), "ms/step", round(j["ms_per_step"], 3), "frac", round(r["frac"], 3), "vs", j.get("vs_baseline"), "exact", (j.get("exact_split_baseline") or {}).get("value"), "sclk", r["board"]["sclk_mhz"], j["config"]["clock_prewarm"]["seconds"])
    except Exception as e:
        print(f, "FAILED", e)
PY
