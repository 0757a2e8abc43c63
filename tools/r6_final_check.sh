#!/bin/bash
# round 6: what the driver runs at round end — the GPU suite, smoke(), the default bench line — on one box
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r6_final_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r6_final_gpu_tests.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r6_final_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r6_final_smoke.log
( time timeout 1500 python bench.py ) > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err
tail -4 gpurun_out/r6_final_gpu_tests.log; tail -3 gpurun_out/r6_final_smoke.log; tail -4 gpurun_out/r6_final_bench.err; head -c 400 gpurun_out/r6_final_bench.json
