#!/bin/bash
# Build diagnostic variants of the conv kernel (K-loop ablations) next to the product library; run with
#   PG_LIB=pose-transfer_amd/lib/libposegan_hip_ab<N>.so python tools/conv_bench.py 4 dec4
set -e
cd "$(dirname "$0")/.."
C=pose-transfer_amd/csrc; L=pose-transfer_amd/lib
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -Iinclude -I$C -DPG_ABLATE=$n -c $C/igemm_conv.hip -o /tmp/igemm_conv_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libposegan_hip_ab$n.so /tmp/igemm_conv_ab$n.o $L/obj/api.o $L/obj/optim.o $L/obj/norm.o $L/obj/losses.o $L/obj/warp.o $L/obj/edge.o $L/obj/small_cin_wgrad.o $L/obj/out_conv_dgrad.o $L/obj/igemm_wgrad.o
done
