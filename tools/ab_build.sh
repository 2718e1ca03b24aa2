#!/bin/bash
# Builds ab/libA.so from HEAD (working-tree changes stashed) and ab/libB.so from the working tree, for same-box A/B
# runs:  gpurun -- 'tools/gpu_ab.sh CJS_HIP_LIB=ab/libA.so CJS_HIP_LIB=ab/libB.so CJS_HIP_LIB=ab/libA.so CJS_HIP_LIB=ab/libB.so'
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
git stash -q -- compressjs-flattened_amd/csrc include
make hip > /dev/null
cp compressjs-flattened_amd/libcjs_hip.so ab/libA.so
git stash pop -q
touch compressjs-flattened_amd/csrc/*.hip
make hip > /dev/null
cp compressjs-flattened_amd/libcjs_hip.so ab/libB.so
ls -la ab/
