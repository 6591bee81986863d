#!/bin/bash
# round 4: the fp16 bulk mode with EIGHT staging waves (768-thread blocks) vs four: bit-equality per layer shape, timings, the mode's tests, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j219; mkdir -p $O
cd $R
echo "== eight staging waves (default)"; timeout 300 python scripts/fp16_stagers_probe.py 2>&1 | grep -v amdgpu | tee $O/probe8.log
echo "== four staging waves (lib_four.so)"; R2DM_HIP_LIB=$R/build_probe/lib_four.so timeout 300 python scripts/fp16_stagers_probe.py 2>&1 | grep -v amdgpu | tee $O/probe4.log
python - <<PY
a=[l.split() for l in open("$O/probe8.log") if "sha1" in l]; b=[l.split() for l in open("$O/probe4.log") if "sha1" in l]
print("bit-identical per shape:", {x[0]: x[2] == y[2] for x, y in zip(a, b)})
PY
timeout 900 python -m pytest tests/test_hip_fp16_mode.py -q 2>&1 | tail -3 | tee $O/pytest_fp16.log
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4 --precision fp16"
for i in 1 2; do
  for m in four eight; do
    if [ $m = four ]; then export R2DM_HIP_LIB=$R/build_probe/lib_four.so; else unset R2DM_HIP_LIB; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 stagers=$m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab_fp16.log
