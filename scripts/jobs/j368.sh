#!/bin/bash
# which neighbour KERNELS make the branch-free fir_up2 wrong?  out_conv (LDS-free, shares CUs) | a 1x1 convolution (LDS, shares CUs) | element-wise torch kernels | (j364: level-1 conv_f16x2, owns its CUs: 0)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j368; mkdir -p $O; cd $R
L=$R/build_probe/lib_vb.so
{ echo "== neighbour out_conv"; R2DM_HIP_LIB=$L NEIGHBOUR=conv HOG_SHAPE=64,2,64,1024,3,8 SECS=10 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep -v amdgpu | cut -c1-300 | tail -3
  echo "== neighbour 1x1 conv 512->512"; R2DM_HIP_LIB=$L NEIGHBOUR=conv HOG_SHAPE=512,512,8,128,1,8 SECS=10 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep -v amdgpu | cut -c1-300 | tail -3
  echo "== neighbour torch element-wise"; R2DM_HIP_LIB=$L NEIGHBOUR=conv HOG_TORCH_ONLY=1 SECS=10 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep -v amdgpu | cut -c1-300 | tail -3
  echo "== neighbour in_conv-like 2->64"; R2DM_HIP_LIB=$L NEIGHBOUR=conv HOG_SHAPE=128,64,64,1024,1,8 SECS=10 timeout 200 python scripts/fir_up_soak.py 2>&1 | grep -v amdgpu | cut -c1-300 | tail -3
} | tee $O/soak.log
