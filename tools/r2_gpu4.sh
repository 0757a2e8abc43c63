#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "transposing or bf16_big or weight_gradient_bf16" --timeout 900 > gpurun_out/r2/t_wtr.log 2>&1; echo "wgrad tr tests rc=$?"; tail -15 gpurun_out/r2/t_wtr.log
python -m pytest tests/test_gpu_round2.py -q -m gpu --timeout 1500 > gpurun_out/r2/tests_round2.log 2>&1; echo "round2 rc=$?"; tail -8 gpurun_out/r2/tests_round2.log
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
PG_ONLY_BF16=1 PG_NO_WGRAD_TR=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --launch-table gpurun_out/r2/lt_bf16_b32.txt > gpurun_out/r2/bench_bf16_b32.json 2>&1; tail -c 3000 gpurun_out/r2/bench_bf16_b32.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_f32_b4.json 2>&1; tail -c 2500 gpurun_out/r2/bench_f32_b4.json
