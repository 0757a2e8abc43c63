#!/bin/bash
# round 6: per-workgroup phase timeline of the quad kernel (timing build) on enc.1 forward / data gradient, batch 32
mkdir -p gpurun_out
export PG_TIMING_EXPERIMENTS=1
for what in fwd dgrad; do
  for q in 1 0; do
    echo "==== enc1 $what PG_BIG_QUAD=$q" >> gpurun_out/r6_quad_timeline.log
    PG_BIG_QUAD=$q PG_DEBUG_CONV_TIMELINE=1 timeout 300 python tools/conv_timeline.py 32 enc1 $what >> gpurun_out/r6_quad_timeline.log 2>&1
  done
done
echo "==== fixed cost: one step only (PG_DEBUG_ONE_KTILE)" >> gpurun_out/r6_quad_timeline.log
PG_DEBUG_ONE_KTILE=1 timeout 300 python tools/layer_bench.py 32 enc1 >> gpurun_out/r6_quad_timeline.log 2>&1
echo "==== no epilogue stores (PG_DEBUG_EPI_NOSTORE)" >> gpurun_out/r6_quad_timeline.log
PG_DEBUG_EPI_NOSTORE=1 timeout 300 python tools/layer_bench.py 32 enc1 >> gpurun_out/r6_quad_timeline.log 2>&1
cat gpurun_out/r6_quad_timeline.log
