#!/bin/bash
# first GPU pass of round 2: new parity tests + the batch-32 bf16 north-star profile (baseline for the kernel work)
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round2.py -q -m gpu -x --timeout 1200 > gpurun_out/r2/tests_round2.log 2>&1
echo "round2 rc=$?" ; tail -15 gpurun_out/r2/tests_round2.log
python -m pytest tests/test_gpu_networks.py -q -m gpu --timeout 1200 > gpurun_out/r2/tests_networks.log 2>&1
echo "networks rc=$?" ; tail -8 gpurun_out/r2/tests_networks.log
OUT=$PWD/gpurun_out/r2/prof_ns0
mkdir -p $OUT
PG_ONLY_BF16=1 rocprofv3 --kernel-trace --stats -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1
grep "generator fwd" $OUT/stdout.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
