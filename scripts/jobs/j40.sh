#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j40; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -s -k "attention" > $O/pytest_att.log 2>&1; grep -a "attention B=\|passed\|failed\|Error" $O/pytest_att.log | tail -20
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-torch-baseline > $O/bench_kt.json 2> $O/bench_kt.err
python - <<PY
import csv, json
j = json.load(open("$O/bench_kt.json")); print("bench", j["value"], j["ms_per_step"])
for r in list(csv.DictReader(open("$O/kt_kernel_stats.csv")))[:12]:
    print("%-80s %6s %9.3f ms %8.1f us" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
