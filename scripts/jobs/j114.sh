#!/bin/bash
# round 3: the whole GPU suite + smoke on the current library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j114; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
