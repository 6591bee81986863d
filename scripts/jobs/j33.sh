#!/bin/bash
# rocprofv3 kernel trace of the bench + a long un-profiled run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j33; mkdir -p $O
cd /tmp
python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-torch-baseline > $O/bench64.json 2> $O/bench64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench_kt -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-torch-baseline > $O/bench_kt.json 2> $O/bench_kt.err
ls -R $O/kt | head -20
python - <<PY
import json, csv, glob
j = json.load(open("$O/bench64.json")); print("64 steps: value", j["value"], "ms/step", j["ms_per_step"], "board", j["roofline"]["board"])
f = glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:25]:
    print("%-100s %6s %10.3f ms %8.1f us %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total kernel ms", tot / 1e6)
PY
