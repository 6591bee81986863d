#!/bin/bash
# is the branch-free fir_up2 itself the victim next to another process?  Its output on fixed inputs, in a loop, beside a convolution loop / a sampler, committed | rewritten body
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j364; mkdir -p $O; cd $R
for lib in default lib_vb; do
  for nb in none conv sampler; do
    if [ $lib = default ]; then NEIGHBOUR=$nb SECS=12 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep fir_up_soak
    else R2DM_HIP_LIB=$R/build_probe/$lib.so NEIGHBOUR=$nb SECS=12 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep fir_up_soak; fi
  done
done | tee $O/soak.log
