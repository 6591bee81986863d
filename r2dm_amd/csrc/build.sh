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
    # Round 5: NO packed-fp32 instruction selection in the kernels that share CUs and whose packed instructions would take SGPR operands (attention, in_conv /
    # out_conv, the FIR resamplers, the posterior): a v_pk_mul / v_pk_fma_f32 with an SGPR-pair source read wrong values in lanes 48-63 whenever the wave shared its
    # CU with an LDS-holding workgroup of ANOTHER PROCESS (profiles/r05_coresidency.txt: the root of the "wrong next to a second process" family).  Costs nothing
    # (attention 77 -> 74 us, out_conv 54 -> 53 us, step +-0: scripts/jobs/j374.sh).  tests/test_host.py checks the built code objects.  (The host pass of hipcc
    # does not know the feature and says so on stderr: filtered.)
    case $f in attention|conv_direct|resample|posterior|norm|embed) extra="$extra -Xclang -target-feature -Xclang -packed-fp32-ops";; esac
    hipcc $FLAGS $extra -c $f.hip -o build/$f.o 2> >(grep -v "packed-fp32-ops' is not a recognized feature" >&2) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
