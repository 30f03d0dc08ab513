#!/bin/bash
# A/B build of the library with extra defines: tools/mk_variant.sh <name> [-DNAME=VALUE ...]  ->  ab/<name>.so
# (used through RT355_LIB=$PWD/ab/<name>.so; *.so is git-ignored, ab/ travels to the GPU box)
name=$1; shift
mkdir -p ab
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-sanitize -Wall -Wno-unused-function \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -munsafe-fp-atomics "$@" -o ab/$name.so raytracer_amd/csrc/rt355.hip
