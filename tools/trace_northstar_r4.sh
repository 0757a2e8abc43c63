# rocprofv3 kernel trace of the north-star pass (generator forward + backward, 256x256, batch 32, bf16 data path) -> timeline text
#   gpurun -- bash tools/trace_northstar_r4.sh [tag]
TAG=${1:-a}
OUT=$PWD/gpurun_out/trace_ns_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PG_ONLY_BF16=1
cd $PWD
PG_NS_ITERS=7 rocprofv3 --kernel-trace -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1 || true
python tools/timeline_r4.py $(ls $OUT/*results.db | head -1) gpurun_out/timeline_ns_$TAG.txt 10
python tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) gpurun_out/kernel_stats_ns_$TAG.csv
rm -rf $OUT
