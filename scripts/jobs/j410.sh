#!/bin/bash
# round 6: whole GPU suite + smoke on the tree with the shared tile end (EPI3), the lane-exchange FIR resamplers and the reversed skip-convolution walk; then a longer alternating A/B of EPI3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j410; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2700 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs"
for i in 1 2 3 4 5 6; do
  for l in build_probe/lib_noepi3.so r2dm_amd/libr2dm_hip.so; do
    R2DM_HIP_LIB=$R/$l timeout 300 python bench.py $A --steps 128 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $l', round(j['ms_per_step'],3), round(j['value'],3), round(j['roofline']['frac'],4))"
  done
done | tee $O/ab.log
