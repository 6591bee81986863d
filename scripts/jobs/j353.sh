#!/bin/bash
# round 5: what makes the neighbour-lane fir_up2 fail the two-rank drop-in test?  The committed library with ONLY fir_up2 replaced by its DPP variant
# (build_probe/lib_dppup.so), and the same with s_nop fences around the two DPP moves (lib_dppupnop.so: an instruction-hazard hypothesis): 8 runs each,
# plus how many of the five files differ when it fails
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j353; mkdir -p $O; cd $R
T="tests/test_dropin_scripts.py::test_sample_and_save_two_ranks_on_one_gpu"
for lib in lib_dppup lib_dppupnop; do
  p=0; for i in 1 2 3 4 5 6 7 8; do R2DM_HIP_LIB=$R/build_probe/$lib.so timeout 600 python -m pytest $T -q -x 2>&1 | grep -Eq "1 passed" && p=$((p+1)); done
  echo "$lib: $p of 8 passed" | tee -a $O/two_ranks.log
done
# single process, the same library: the drop-in script twice, files compared
for lib in lib_dppup; do
  for k in a b; do R2DM_HIP_LIB=$R/build_probe/$lib.so python sample_and_save.py --ckpt /tmp/pytest-of-root/pytest-0/ckpt0/synthetic.pth --output_dir /tmp/one_$k --batch_size 2 --num_samples 5 --num_steps 2 > /dev/null 2>&1; done
  python - <<'PY' | tee -a $O/two_ranks.log
import torch, glob, os
d = 0
for f in sorted(glob.glob('/tmp/one_a/samples_*.pth')):
    a, b = torch.load(f), torch.load(f.replace('one_a', 'one_b'))
    d += int(not torch.equal(a, b))
print("single process, two runs of the script with lib_dppup: files differing", d)
PY
done
