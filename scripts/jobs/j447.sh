#!/bin/bash
# final check of the fp16-input out_conv (two channels per load batch): fp16-mode tests, fp16 bench A/B against the library before it, fp32 bench unchanged
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/j447; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_fp16_mode.py tests/test_hip_kernels.py tests/test_hip_configs.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4; do
  for l in head2 fin; do
    R2DM_HIP_LIB=$R/build_probe/lib_$l.so timeout 300 python bench.py $A --precision fp16 --steps 128 --warmup 4 2>$O/err_$l.log | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 lib=$l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
timeout 600 python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err; python -c "
import json; j=json.loads(open('$O/bench_fp16.json').read().strip().splitlines()[-1]); print('fp16 default line', j['value'], j['ms_per_step'], j['vs_torch_rocm_baseline'], j['torch_rocm_baseline']['compiled_fp16_autocast']['value'])" | tee -a $O/ab.log
