#!/bin/bash
# round 5: the new tests (second golden resolution, FIR fixture, batch-8 sampler vs fp64, late range trip, two-rank configs incl. 128x2048, layout hash) and the new bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j305; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -x -m gpu -k "second_golden or batch8_full_size_sampler or fir_golden or late_range or two_rank or adopted or rccl or falls_back or really_leave" -s > $O/pytest_new.log 2>&1; tail -6 $O/pytest_new.log; grep "batch-8 sampler" $O/pytest_new.log
cd /tmp
timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -3 $O/bench_n1.err
python - <<PY
import json
j=json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print(round(j["value"],3), round(j["ms_per_step"],3), j["roofline"]["frac"], j["vs_baseline"], j["rccl"], j["arithmetic"])
print({k:(round(v["value"],3) if isinstance(v,dict) and "value" in v else v) for k,v in j["torch_rocm_baseline"].items() if k in ("value","fp16_autocast","compiled_fp16_autocast")})
print(j["torch_rocm_baseline"].get("compiled_fp16_autocast"))
print(j["cpu_baseline"])
PY
