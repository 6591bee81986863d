#!/bin/bash
# round 4: the wide tile's biases staged through LDS by the stagers (with the (a, d) table) instead of 16 global loads at the tile's end: bit-equality, timeline, step A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j225; mkdir -p $O
cd $R
for pc in 2 1; do
  PIECES=$pc COT=128 timeout 300 python scripts/fp16_stagers_probe.py 2>&1 | grep -v amdgpu > $O/probe_lib_$pc.log
  PIECES=$pc COT=128 R2DM_HIP_LIB=$R/build_probe/lib_biaslds.so timeout 300 python scripts/fp16_stagers_probe.py 2>&1 | grep -v amdgpu > $O/probe_new_$pc.log
  python - <<PY
a=[l.split() for l in open("$O/probe_lib_$pc.log") if "sha1" in l]; b=[l.split() for l in open("$O/probe_new_$pc.log") if "sha1" in l]
print("pieces $pc bit-identical per shape:", {x[0]: x[2] == y[2] for x, y in zip(a, b)})
print("us lib:", [x[-6] for x in a]); print("us new:", [x[-6] for x in b])
PY
done
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2 3; do
  for m in lib biaslds; do
    if [ $m = biaslds ]; then export R2DM_HIP_LIB=$R/build_probe/lib_biaslds.so; else unset R2DM_HIP_LIB; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab.log
unset R2DM_HIP_LIB
cd $R; R2DM_HIP_LIB=$R/build_probe/lib_biaslds.so timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py -q -x 2>&1 | tail -2
