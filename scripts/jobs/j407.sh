#!/bin/bash
# round 6: FIR kernels after the up-sampler's chain was unified (a sample's bits must not depend on the batch it is part of), then the stagger timelines (j405)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j407; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x -k "fir or unet or north_star or batch8 or two_rank or configs or golden" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
bash $R/scripts/jobs/j405.sh
