#!/bin/bash
# Diagnostic build: the library with ONE translation unit recompiled with extra flags -> trace_tmp/libess_variant.so
# usage: tools/build_variant.sh <file stem, e.g. conv_bf16_wide> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
stem=$1; shift
mkdir -p trace_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -x hip -c ess_amd/csrc/$stem.hip -o trace_tmp/$stem.var.o
objs=$(ls ess_amd/csrc/_build/*.o | grep -v "/$stem.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs trace_tmp/$stem.var.o -o trace_tmp/libess_variant.so
echo built trace_tmp/libess_variant.so
