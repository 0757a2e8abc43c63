set -e
OUT=$PWD/gpurun_out/prof_ns
mkdir -p $OUT
export TMPDIR=/tmp
PG_ONLY_BF16=1 rocprofv3 --kernel-trace --stats -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1 || true
tail -1 $OUT/stdout.log | cut -c1-200
