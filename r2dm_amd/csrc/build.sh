#!/bin/bash
# Builds libr2dm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libr2dm_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm"
mkdir -p build
pids=()
for f in conv_mfma conv_bf16x3 conv_f16x2 proj_f16x2 presplit conv_direct norm resample attention embed posterior engine; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ conv_epilogue.h -nt build/$f.o ] || [ conv_bf16x3.h -nt build/$f.o ] || [ f16x2.h -nt build/$f.o ] || [ wave_ops.h -nt build/$f.o ] || [ gn_math.h -nt build/$f.o ] || [ ../../include/r2dm_hip.h -nt build/$f.o ]; then
    extra=""; case $f in conv_bf16x3*|conv_f16x2|proj_f16x2|presplit) extra="-fno-slp-vectorize";; esac  # packed f32 VALU next to MFMAs is an anti-lever
    hipcc $FLAGS $extra -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
