#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "warp or transposing or output_conv or bf16" --timeout 900 > gpurun_out/r2/t_k.log 2>&1; echo "kernel tests rc=$?"; tail -12 gpurun_out/r2/t_k.log
python -m pytest tests/test_gpu_round2.py tests/test_gpu_networks.py -q -m gpu --timeout 1500 > gpurun_out/r2/tests_net.log 2>&1; echo "networks+round2 rc=$?"; tail -8 gpurun_out/r2/tests_net.log
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --launch-table gpurun_out/r2/lt_bf16_b32.txt > gpurun_out/r2/bench_bf16_b32.json 2>&1; tail -c 3300 gpurun_out/r2/bench_bf16_b32.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_f32_b4.json 2>&1; tail -c 2600 gpurun_out/r2/bench_f32_b4.json | head -c 700
OUT=$PWD/gpurun_out/r2/prof_ns2
mkdir -p $OUT
PG_NO_SIDE_STREAM=1 PG_ONLY_BF16=1 rocprofv3 --kernel-trace --stats -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1
grep "generator fwd" $OUT/stdout.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
