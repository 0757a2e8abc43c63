#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench workload; run on the GPU box through gpurun:
#   gpurun -- bash tools/profile_bench.sh <tag>
# Writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv summary into profiles/.
set -e
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/bench_stdout.log 2>&1 || true
tail -2 $OUT/bench_stdout.log
find $OUT -name "*stats*" | head
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -40 "$F"
# keep the merge small: drop the per-dispatch trace if it is large
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
