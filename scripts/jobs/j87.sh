#!/bin/bash
# round 3: ablation timings of conv_f16x2 (wrong results, timing only) + in-kernel timeline of the current code + default bench with other_configs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j87; mkdir -p $O
cd $R
for lib in r2dm_amd/libr2dm_hip.so build_probe/lib_noxf.so build_probe/lib_nomfma.so build_probe/lib_nodma.so build_probe/lib_noxf_nodma.so; do
  echo "== $lib"; R2DM_HIP_LIB=$R/$lib SHAPES=L1_64_64,L2_128_128,L4_512_512,L1_64_128 ITERS=30 timeout 200 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
done | tee $O/ablation.log
B=8 R2DM_HIP_LIB=$R/build_probe/lib_f2_prof.so MAXEV=260 SHAPES=L1_64_64 timeout 300 python scripts/f2_timeline.py 2>&1 | grep -v amdgpu > $O/tl.log; sed -n 1,2p $O/tl.log; sed -n 60,200p $O/tl.log
cd /tmp; python $R/bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; python -c "
import json; j=json.load(open('$O/bench_n1.json')); print(round(j['value'],3), j.get('vs_baseline'), j.get('other_configs'))"
