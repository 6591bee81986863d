#!/bin/bash
# round 3, weak #1: aggressor / victim.  j75/j76: the failure disappears with commit 1c7a0dc, which ONLY replaces the out_conv
# kernel (conv_direct_kernel<2>: 12.9 KB static + 4.6 KB dynamic LDS) by an LDS-free one.  bad = tree d5e0cd1 (fails 111/150,
# bf16x3 300/300), good = tree 1c7a0dc / HEAD.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j77; mkdir -p $O
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | grep -v amdgpu.ids
BAD=$R/build_probe/bis_d5e0cd1; GOOD=$R/build_probe/bis_1c7a0dc; P=$R/build_probe/cwsr_lds_probe
run() { (cd $1 && shift && env "$@" timeout 400 python $R/scripts/stress_cross.py 2>&1 | grep -v amdgpu.ids | grep -E "forwards differ|Error|error" ); }
{
echo "== 1. victim = good tree, neighbour = bad tree's forwards"
run $GOOD TAG=good_vs_badfwd HOG_CMD="cd $BAD && ROLE=hog SECS=200 python scripts/stress_shared_forward.py"
echo "== 2. victim = bad tree, neighbour = good tree's forwards"
run $BAD TAG=bad_vs_goodfwd HOG_CMD="cd $GOOD && ROLE=hog SECS=200 python scripts/stress_shared_forward.py"
echo "== 3. victim = HEAD, neighbour = ONLY the old out_conv kernel (bad tree, 64 -> 2 at 64x1024, batch 8) in a loop"
run $R TAG=head_vs_oldoutconv HOG_WAIT=25 HOG_CMD="cd $BAD && SECS=200 python $R/scripts/hog_conv_loop.py"
echo "== 4. victim = HEAD, neighbour = the LDS-free out_conv kernel of HEAD in a loop (control)"
run $R TAG=head_vs_newoutconv HOG_WAIT=25 HOG_CMD="cd $R && SECS=200 python $R/scripts/hog_conv_loop.py"
echo "== 5. victim = bad tree, neighbour = none but ITS OWN forwards are all there is (control: alone)"
run $BAD TAG=bad_alone
echo "== 6. victim = bad tree, neighbour = HEAD's LDS-free out_conv kernel in a loop (any neighbour enough?)"
run $BAD TAG=bad_vs_newoutconv HOG_WAIT=25 HOG_CMD="cd $R && SECS=200 python $R/scripts/hog_conv_loop.py"
echo "== 7. victim = bad tree, neighbour = a 1x1 GEMM-like convolution of HEAD (fp16 pipe, 55 KB LDS) in a loop"
run $BAD TAG=bad_vs_proj HOG_WAIT=25 HOG_CMD="cd $R && SHAPE=512,512,8,128,1,8 SECS=200 python $R/scripts/hog_conv_loop.py"
echo "== 8. LDS integrity probe (158 KB / 60 KB of LDS per block, idle blocks) next to the old out_conv kernel in a loop"
(cd $BAD && SECS=60 python $R/scripts/hog_conv_loop.py > /dev/null 2>&1 &) ; sleep 25
timeout 100 $P 161792 300 300 256 0; timeout 100 $P 61440 300 300 256 0; timeout 100 $P 17472 300 300 2048 0
sleep 20
} 2>&1 | grep -v amdgpu.ids | tee $O/cross.log
