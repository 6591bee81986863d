#!/bin/bash
# round 6: the hygiene batch on the GPU -- whole GPU suite (incl. the new tests: bench --gpus 2 self-launch, schedule_on="device", checkpoint pre-flight, per-site range report) + smoke + the pre-flight CLI on the synthetic checkpoint
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j402; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -x -k "preflight or schedule_scalars_on_the_device or launches_its_own_ranks or range" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 300 python -m r2dm_amd.check synthetic --steps 32 --batch 2 --top 12 2>&1 | grep -v amdgpu | tee $O/check_synthetic.txt | tail -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
