#!/bin/bash
# build_probe/lib_<name>.so = libr2dm_hip.so with every kernel source compiled with extra -D flags (ablation
# experiments; select with R2DM_HIP_LIB=build_probe/lib_<name>.so).  Usage: scripts/build_variant.sh <name> [-DFLAG ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../r2dm_amd/csrc"
out=../../build_probe/obj_$name
mkdir -p $out
pids=()
for f in conv_mfma conv_bf16x3 conv_f16x2 proj_f16x2 presplit conv_direct norm resample attention embed posterior engine; do
  extra=""; case $f in conv_bf16x3*|conv_f16x2|proj_f16x2|presplit) extra="-fno-slp-vectorize";; esac
  case $f in attention|conv_direct|resample|posterior|norm|embed) extra="$extra -Xclang -target-feature -Xclang -packed-fp32-ops";; esac  # (as r2dm_amd/csrc/build.sh)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra "$@" -c $f.hip -o $out/$f.o 2> $out/$f.err &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed:"; head -5 $out/*.err; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o ../../build_probe/lib_$name.so
echo built lib_$name.so
