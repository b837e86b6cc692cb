#!/bin/bash
# usage: tools/build_variant.sh <file.hip> <out.so> [-DFLAG ...]
# Builds a copy of librdfc_hip.so in which one source is recompiled with extra flags (for interleaved A/B runs
# through RDFC_HIP_LIB on the GPU box).  Run `python -m rdfc_gan_amd.build` first so every other object exists.
src=$1; out=$2; shift 2
B=rdfc_gan_amd/csrc/build
mkdir -p $(dirname $out)
obj=$(dirname $out)/$(basename $out .so)_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-value -Wno-pass-failed "$@" -c rdfc_gan_amd/csrc/$src -o $obj || exit 1
objs=$(ls $B/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs $obj
