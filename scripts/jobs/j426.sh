#!/bin/bash
# round 6: range slots as running maxima (reset only on a trip): range / check / unet tests, then sites vs shared slot again (j425)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j426; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_range.py tests/test_hip_unet.py tests/test_hip_configs.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
JOBDIR=j426 bash -c "sed -e 's#gpurun_out/j425#gpurun_out/j426#' $R/scripts/jobs/j425.sh > /tmp/j425b.sh; bash /tmp/j425b.sh"
