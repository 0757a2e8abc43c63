# single-stream rocprofv3 kernel statistics of the north-star pass: every kernel's duration is its own (no co-running kernels)
TAG=${1:-ss}
OUT=$PWD/gpurun_out/trace_ns_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PG_ONLY_BF16=1 PG_NO_SIDE_STREAM=1
PG_NS_ITERS=7 rocprofv3 --kernel-trace -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1 || true
python tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) gpurun_out/kernel_stats_ns_$TAG.csv
python tools/timeline_r4.py $(ls $OUT/*results.db | head -1) gpurun_out/timeline_ns_$TAG.txt 10
rm -rf $OUT
