#!/bin/bash
# full GPU suite + smoke + bench in both operand-split modes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j29; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tee $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python bench.py --steps 20 --warmup 5 --precision fp32-bf16x3 --no-cpu-baseline --no-torch-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
python - <<PY
import json
for f in ("bench.json", "bench_bf16x3.json"):
    j = json.load(open("$O/" + f)); r = j["roofline"]
    print(f, "value", round(j["value"], 3), "ms/step", round(j["ms_per_step"], 3), "frac", round(r["frac"], 3), "achieved", round(r["achieved"], 1), "peak", round(r["peak"], 1),
          "board", r["board"], "frac@clk", r.get("frac_at_sustained_clock"))
    for e in [r["dominant_kernel"]] + r["other_conv_kernels"]:
        print("   ", e["kernel"], "launches/step", e["launches_per_step"], "ms/step", round(e["ms_per_step"], 3), "TF/s", round(e["tflops"], 1), "frac", round(e["frac"], 3), "share", round(e["share_of_conv_flops"], 3))
    print("    torch", j.get("torch_rocm_baseline"), "cpu", j.get("cpu_baseline", {}).get("value"), "hipblaslt", r.get("hipblaslt_bf16_gemm"))
PY
