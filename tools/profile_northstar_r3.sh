# rocprofv3 kernel statistics of the north-star sub-metric (generator forward + backward, 256x256, batch 32, bf16 data path)
#   gpurun -- bash tools/profile_northstar_r3.sh [tag]     -> gpurun_out/round3_kernel_stats_northstar_<tag>.csv
TAG=${1:-bf16}
OUT=$PWD/gpurun_out/prof_ns_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PG_ONLY_BF16=1
python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd" | tee gpurun_out/round3_northstar_$TAG.txt
PG_NS_ITERS=10 rocprofv3 --kernel-trace --stats -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1 || true
python tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) gpurun_out/round3_kernel_stats_northstar_$TAG.csv
PG_NO_SIDE_STREAM=1 PG_NS_ITERS=10 rocprofv3 --kernel-trace --stats -d $OUT/ss -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout_ss.log 2>&1 || true
python tools/rocpd_summary.py $(ls $OUT/ss/*results.db | head -1) gpurun_out/round3_kernel_stats_northstar_${TAG}_single_stream.csv
rm -rf $OUT
