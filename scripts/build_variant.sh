#!/bin/bash
# build_probe/lib_<name>.so = libr2dm_hip.so with conv_bf16x3.hip compiled with extra -D flags (ablation experiments)
set -e
name=$1; shift
cd "$(dirname "$0")/../r2dm_amd/csrc"
mkdir -p ../../build_probe/obj_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c conv_bf16x3.hip -o ../../build_probe/obj_$name/conv_bf16x3.o 2>/dev/null
objs=$(ls build/*.o | grep -v conv_bf16x3.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../build_probe/obj_$name/conv_bf16x3.o -o ../../build_probe/lib_$name.so
echo built lib_$name.so
