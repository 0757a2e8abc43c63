#!/bin/bash
# round 6: the tap-quad kernel — parity tests, then enc.1 / enc.2 / dec.5 in isolation with and without it (batch 32)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "quad" > gpurun_out/r6_quad_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_quad_tests.log
for q in 1 0; do
  echo "---- PG_BIG_QUAD=$q" >> gpurun_out/r6_quad_layers.log
  PG_BIG_QUAD=$q timeout 600 python tools/layer_bench.py 32 enc1 enc2 dec5 >> gpurun_out/r6_quad_layers.log 2>&1
done
tail -5 gpurun_out/r6_quad_tests.log; cat gpurun_out/r6_quad_layers.log
