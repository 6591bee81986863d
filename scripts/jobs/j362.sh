#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do timeout 900 python -m pytest tests/test_dropin_scripts.py -q -k two_ranks 2>&1 | grep -v amdgpu | tail -1; done
