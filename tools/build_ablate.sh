#!/bin/bash
# Diagnostic build: the library with the ablation switches of the 3x3 kernels compiled in (-DESS_ABLATE; ESS_WS_ABL=<bits> at run
# time: 2 = no fragment reads / MFMAs, 4 = no global loads, 8 = no epilogue, 16 = no LDS writes) -> trace_tmp/libess_ablate.so.
# Never the shipped library.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DESS_ABLATE -x hip"
for f in conv_bf16_wide conv_bf16_ws conv_fwd; do
  /opt/rocm/bin/hipcc $F -c ess_amd/csrc/$f.hip -o trace_tmp/$f.abl.o &
done
wait
objs=$(ls ess_amd/csrc/_build/*.o | grep -v -e conv_bf16_wide -e conv_bf16_ws -e conv_fwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs trace_tmp/conv_bf16_wide.abl.o trace_tmp/conv_bf16_ws.abl.o trace_tmp/conv_fwd.abl.o -o trace_tmp/libess_ablate.so
echo built trace_tmp/libess_ablate.so
