#!/bin/bash
# round 5, final code (a conv_f16x2 block owns its CU): the neighbour-process test 8 times, 20 probe runs of 600 fp16-mode forwards next to
# the strongest neighbour, then the evidence set (scripts/jobs/j306.sh)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j331; mkdir -p $O; cd $R
p=0; for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_hip_configs.py -q -x -k "next_to_a_second" 2>&1 | grep -Eq "[0-9]+ passed" && p=$((p+1)); done; echo "neighbour test runs without an abort: $p of 8" | tee $O/neighbour.log
f=0; for i in $(seq 1 20); do MODES=fp16 HOG_SHAPE=64,2,64,1024,3,8 REPS=600 MODE=process timeout 120 python scripts/coresidency_probe.py 2>&1 | grep -q "Memory access" && f=$((f+1)); done; echo "faults $f of 20 (fp16 mode, out_conv neighbour, final library)" | tee -a $O/neighbour.log
JOB=j331 bash scripts/jobs/j306.sh
