#!/bin/bash
# round 5, third session: step A/B of the small-kernel changes (base = e7bcddb + the two new C-ABI symbols as stubs | new), alternating in one job,
# then the r05c evidence set on the final code (scripts/jobs/j306.sh)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j344; mkdir -p $O; cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4 5; do
  R2DM_HIP_LIB=$R/build_probe/lib_base.so python $R/bench.py $A --prewarm-s 1.0 2>$O/base.err | line base
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line new
done | tee $O/ab.log
JOB=j344 bash $R/scripts/jobs/j306.sh
