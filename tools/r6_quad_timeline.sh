#!/bin/bash
# round 6: the tap-quad kernels on encoder level 1 at batch 32 — isolated launch times of the three arms, then the per-workgroup phase
# timeline (timing build: PG_TIMING_EXPERIMENTS=1 python pose-transfer_amd/runtime/build.py) of the 8-wave and the 4-wave form
#   -> profiles/round6_quad_layers.txt, round6_quad_timeline_enc1_{8,4}wave.txt
mkdir -p gpurun_out
rm -f gpurun_out/r6_quad_layers.log gpurun_out/r6_quad_timeline.log
for cfg in "PG_BIG_QUAD=1 PG_QUAD_WAVES=8" "PG_BIG_QUAD=1 PG_QUAD_WAVES=4" "PG_BIG_QUAD=0"; do
  echo "---- $cfg" >> gpurun_out/r6_quad_layers.log
  env $cfg timeout 600 python tools/layer_bench.py 32 enc1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_quad_layers.log
done
export PG_TIMING_EXPERIMENTS=1
for w in 8 4; do
  for what in fwd dgrad; do
    echo "==== enc1 $what, $w waves" >> gpurun_out/r6_quad_timeline.log
    PG_QUAD_WAVES=$w PG_TL_DISPATCH=1 PG_DEBUG_CONV_TIMELINE=1 timeout 300 python tools/conv_timeline.py 32 enc1 $what 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_quad_timeline.log
  done
done
echo "==== fixed cost: one step only (PG_DEBUG_ONE_KTILE) / no epilogue stores (PG_DEBUG_EPI_NOSTORE)" >> gpurun_out/r6_quad_timeline.log
PG_DEBUG_ONE_KTILE=1 timeout 300 python tools/layer_bench.py 32 enc1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_quad_timeline.log
PG_DEBUG_EPI_NOSTORE=1 timeout 300 python tools/layer_bench.py 32 enc1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_quad_timeline.log
cat gpurun_out/r6_quad_layers.log gpurun_out/r6_quad_timeline.log
