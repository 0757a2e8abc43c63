#!/bin/bash
# round 6: new tests (main.py loop, bf16 warp vs oracle, deterministic wide warp), the cosine / relative-L2 study, the default bench line
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -k "not quad" > gpurun_out/r6_tests2.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_tests2.log
PG_TOL_STUDY=1 timeout 1500 python -m pytest tests/test_gpu_round5.py -q -s -k "test_bf16_data_step_256_vs_reference and (93 or 94)" 2>&1 | grep -E "TOLSTUDY|passed|failed" > gpurun_out/r6_cosine_study.log
( time timeout 1500 python bench.py ) > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -30 gpurun_out/r6_tests2.log; cat gpurun_out/r6_cosine_study.log; tail -5 gpurun_out/r6_bench_default.err; head -c 600 gpurun_out/r6_bench_default.json
