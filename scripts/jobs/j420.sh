#!/bin/bash
# round 6 evidence set (r06a) on the round's code: rocprofv3 kernel trace + three PMC passes + bench lines + smoke + whole GPU suite + 256-step validation (scripts/jobs/j306.sh)
JOB=j420 bash $GRAFT_REPO_ROOT/scripts/jobs/j306.sh
