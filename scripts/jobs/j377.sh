#!/bin/bash
# round 5, last session: the r05d evidence set on the round's FINAL code (after the packed-fp32 build change of attention / in_conv / out_conv / FIR / posterior):
# rocprofv3 kernel trace + three PMC passes + bench lines + smoke + whole GPU suite + 256-step validation (scripts/jobs/j306.sh)
JOB=j377 bash $GRAFT_REPO_ROOT/scripts/jobs/j306.sh
