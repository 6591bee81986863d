#!/bin/bash
# round 3, weak #1: what exactly is lost when a second process shares the GPU?  (a) platform probe: LDS / register integrity of
# idle blocks next to a forward-running neighbour process, by LDS size; (b) the forward stress with bisect variants.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j73; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
P=$R/build_probe/cwsr_lds_probe
{
echo "== control: no neighbour"
timeout 120 $P 161792 200 300 256 0
timeout 120 $P 161792 200 300 256 1
echo "== neighbour process: U-Net forwards (default kernels)"
ROLE=hog SECS=170 python scripts/stress_shared_forward.py &
HOG=$!
sleep 45
for sz in 16384 61440 69632 102400 161792; do timeout 120 $P $sz 400 300 256 0; done
timeout 120 $P 161792 400 300 256 1
timeout 120 $P 161792 400 300 256 2
timeout 120 $P 161792 100 2000 256 0
timeout 120 $P 32768 400 300 1024 0
wait $HOG
} 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
{
TAG=default timeout 300 python scripts/stress_shared_forward.py
TAG=nonpersistent R2DM_F2_NONPERSISTENT=1 HOG_ENV="R2DM_F2_NONPERSISTENT=1" timeout 300 python scripts/stress_shared_forward.py
TAG=hog_f32 HOG_ENV="R2DM_CONV_ALGO=f32" timeout 300 python scripts/stress_shared_forward.py
TAG=cumask HSA_CU_MASK=0:0-127 HOG_ENV="HSA_CU_MASK=0:128-255" timeout 300 python scripts/stress_shared_forward.py
} 2>&1 | grep -v amdgpu.ids | tee $O/stress_forward.log
HOGS=forward ITERS=400 timeout 400 python scripts/stress_shared_gpu.py 2>&1 | grep -v amdgpu.ids | tee $O/stress_conv.log
