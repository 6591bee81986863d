#!/bin/bash
# build_probe/bis_<commit> = the tree of <commit> with its own libr2dm_hip.so (shared-GPU bisect, scripts/jobs/j75.sh)
set -e
cd "$(dirname "$0")/.."
for c in "$@"; do
  d=build_probe/bis_$c
  rm -rf $d; mkdir -p $d
  git archive $c | tar -x -C $d
  (cd $d/r2dm_amd/csrc && ./build.sh > /dev/null 2>&1 && echo "built $c")
  rm -rf $d/tests/golden $d/profiles $d/r2dm_amd/csrc/build $d/gpurun_out
done
