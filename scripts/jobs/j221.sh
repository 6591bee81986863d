#!/bin/bash
# round 4: the stagers' transform over the whole quad in one pass (32 values per stage, -DF2_XF_FULL) vs two halves in two segments
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j221; mkdir -p $O
cd $R
for pc in 2 1; do
  echo "== pieces $pc, library"; PIECES=$pc timeout 300 python scripts/fp16_stagers_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_lib_$pc.log
  echo "== pieces $pc, one-pass transform"; PIECES=$pc R2DM_HIP_LIB=$R/build_probe/lib_xffull.so timeout 300 python scripts/fp16_stagers_probe.py 2>&1 | grep -v amdgpu | tee $O/probe_xf_$pc.log
  python - <<PY
a=[l.split() for l in open("$O/probe_lib_$pc.log") if "sha1" in l]; b=[l.split() for l in open("$O/probe_xf_$pc.log") if "sha1" in l]
print("bit-identical per shape:", {x[0]: x[2] == y[2] for x, y in zip(a, b)})
PY
done
cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --steps 64 --warmup 4"
for i in 1 2 3; do
  for m in lib xffull; do
    if [ $m = xffull ]; then export R2DM_HIP_LIB=$R/build_probe/lib_xffull.so; else unset R2DM_HIP_LIB; fi
    timeout 300 python $R/bench.py $A 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
  done
done | tee $O/ab.log
for m in lib xffull; do
    if [ $m = xffull ]; then export R2DM_HIP_LIB=$R/build_probe/lib_xffull.so; else unset R2DM_HIP_LIB; fi
    timeout 300 python $R/bench.py $A --precision fp16 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp16 $m', round(j['ms_per_step'],3), round(j['value'],3), round(j.get('roofline',{}).get('frac'),4))"
done | tee -a $O/ab.log
