set -e
OUT=$PWD/gpurun_out/prof_bf
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --precision bf16_data > $OUT/bench_stdout.log 2>&1 || true
tail -1 $OUT/bench_stdout.log | cut -c1-160
