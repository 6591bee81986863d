#!/bin/bash
# round 5: conv_f16x2's output stores (and residual loads) with the non-temporal hint -- alternating bench lines default | nt | nt + nt residual, one job
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j346; mkdir -p $O; cd /tmp
A="--no-cpu-baseline --no-torch-baseline --no-exact-baseline --no-other-configs --no-compile-baseline"
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step', round(j['value'],3), 'images/s', 'conv us', round(j['roofline']['dominant_kernel']['avg_launch_us'],2))"; }
for i in 1 2 3 4; do
  python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line default
  R2DM_HIP_LIB=$R/build_probe/lib_ntx.so python $R/bench.py $A --prewarm-s 1.0 2>/dev/null | line nt_xload
done | tee $O/ab.log
