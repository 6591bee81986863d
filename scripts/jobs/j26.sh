#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j26; mkdir -p $O
cd $R
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -30
cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap 2>/dev/null
{
WORK=idle timeout 60 python scripts/power_probe.py
WORK=gemm_bf16 timeout 60 python scripts/power_probe.py
for w in L1_64_64 L2_128_128 L3_256_256; do
  R2DM_SPEC=0 WORK=$w timeout 60 python scripts/power_probe.py
  R2DM_SPEC=1 WORK=$w timeout 60 python scripts/power_probe.py
  R2DM_CONV_ALGO=f32 WORK=$w timeout 60 python scripts/power_probe.py
done
} 2>&1 | grep -v amdgpu.ids | tee $O/power.log
