#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16" --timeout 900 > gpurun_out/r2/t_big.log 2>&1; echo "bf16 kernel tests rc=$?"; tail -12 gpurun_out/r2/t_big.log
python -m pytest "tests/test_gpu_networks.py" -q -m gpu -k "edge_shapes or bf16 or operand" --timeout 900 > gpurun_out/r2/t_edge.log 2>&1; echo "networks subset rc=$?"; tail -8 gpurun_out/r2/t_edge.log
python tools/dbg_dp_identity.py > gpurun_out/r2/dbg_dp.log 2>&1; cat gpurun_out/r2/dbg_dp.log | head -60
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
PG_ONLY_BF16=1 PG_NO_BF16_BIG=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --launch-table gpurun_out/r2/lt_bf16_b32.txt > gpurun_out/r2/bench_bf16_b32.json 2>&1; tail -c 600 gpurun_out/r2/bench_bf16_b32.json
OUT=$PWD/gpurun_out/r2/prof_ns1
mkdir -p $OUT
PG_NO_SIDE_STREAM=1 PG_ONLY_BF16=1 rocprofv3 --kernel-trace --stats -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1
grep "generator fwd" $OUT/stdout.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
