#!/bin/bash
# round 6: in-process A/B of the weight-gradient workgroup targets on the north-star pass and on the batch-32 step (PG_DYN_ENV=1)
mkdir -p gpurun_out; rm -f gpurun_out/r6_wgtr_ab.log
export PG_DYN_ENV=1
A1="t128:PG_WGTR_TARGET=128;t160:PG_WGTR_TARGET=160;t192:PG_WGTR_TARGET=192;t256:PG_WGTR_TARGET=256"
A2="q256:PG_WGTR4_TARGET=256;q192:PG_WGTR4_TARGET=192;q320:PG_WGTR4_TARGET=320;q512:PG_WGTR4_TARGET=512"
for arms in "$A1" "$A2"; do
  echo "==== pass: $arms" >> gpurun_out/r6_wgtr_ab.log
  PG_AB_ARMS="$arms" python tools/quad_inproc_ab.py 6 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_wgtr_ab.log
  echo "==== step (batch 32): $arms" >> gpurun_out/r6_wgtr_ab.log
  PG_AB_MODE=step PG_AB_ARMS="$arms" python tools/quad_inproc_ab.py 6 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_wgtr_ab.log
done
cat gpurun_out/r6_wgtr_ab.log
