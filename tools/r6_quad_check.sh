#!/bin/bash
# round 6: the tap-quad kernels — parity tests, then enc.1 in isolation: 4-wave form (two workgroups per CU), 8-wave form, tap-pair kernels
mkdir -p gpurun_out
rm -f gpurun_out/r6_quad_layers.log
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "quad" > gpurun_out/r6_quad_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_quad_tests.log
for cfg in "PG_BIG_QUAD=1 PG_QUAD_WAVES=4" "PG_BIG_QUAD=1 PG_QUAD_WAVES=8" "PG_BIG_QUAD=0"; do
  echo "---- $cfg" >> gpurun_out/r6_quad_layers.log
  env $cfg timeout 600 python tools/layer_bench.py 32 enc1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_quad_layers.log
done
export PG_TIMING_EXPERIMENTS=1
rm -f gpurun_out/r6_quad_timeline.log
for what in fwd dgrad; do
  echo "==== enc1 $what 4-wave" >> gpurun_out/r6_quad_timeline.log
  PG_DEBUG_CONV_TIMELINE=1 timeout 300 python tools/conv_timeline.py 32 enc1 $what 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_quad_timeline.log
done
tail -5 gpurun_out/r6_quad_tests.log; cat gpurun_out/r6_quad_layers.log gpurun_out/r6_quad_timeline.log
